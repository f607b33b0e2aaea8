/* libuvx — C ABI of the MI355X-native Ultravox audio->LLM hot path (gfx950 / CDNA4).
 *
 * The reference (fixie-ai/ultravox) has NO native/FFI layer: its hot path is reached through Python
 * classes (ultravox/model/ultravox_model.py:277-352 UltravoxModel.forward, :354-396
 * _prepare_audio_embeds, :768-800 UltravoxProjector.forward, :865-994 ModifiedWhisperEncoder.forward;
 * ultravox/model/ultravox_processing.py:217-370 UltravoxProcessor.__call__).  This header is the
 * boundary a replacement exports instead; each entry point names the reference code it replaces.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer to a contiguous, caller-owned buffer (row-major layouts
 *    exactly as documented per argument); the library never allocates or frees device memory;
 *  - `stream` is a hipStream_t; every call is asynchronous on it, no hidden synchronisation;
 *  - scratch comes from a caller-allocated workspace; size it with the *_ws_bytes functions;
 *  - return value: 0 ok, <0 uvx error (below), >0 a hipError_t; message via uvx_last_error()
 *    (thread local); no exception crosses the boundary;
 *  - dtype: UVX_BF16 (production: bf16 storage, f32 accumulate/statistics) or UVX_F32 (parity mode,
 *    every tensor f32);
 *  - threading: one process per GPU; a handle-free API.  The only process-wide mutable state is the set of tuning knobs
 *    (uvx_set_option, uvx_gemm_force_variant, uvx_gemm_override_variant, uvx_attention_force_qt, uvx_prof_*): set them
 *    before the first compute call, not concurrently with one; everything else is re-entrant.
 */
#ifndef UVX_H_
#define UVX_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define UVX_ABI_VERSION 19
#define UVX_BF16 0
#define UVX_F32 1

#define UVX_OK 0
#define UVX_ERR_INVALID (-1)     /* bad argument                       -> ValueError in the Python mirror */
#define UVX_ERR_SHAPE (-2)       /* unsupported / inconsistent shape   -> ValueError */
#define UVX_ERR_WORKSPACE (-3)   /* workspace too small                -> RuntimeError */
#define UVX_ERR_UNSUPPORTED (-4) /* feature not built                  -> RuntimeError */
#define UVX_ERR_RUNTIME (-5)     /* a collective (RCCL) call failed    -> RuntimeError */

const char* uvx_last_error(void);
int32_t uvx_abi_version(void);

/* ============================== model description ============================== */

typedef struct {
  int32_t dtype;
  /* audio tower: Whisper encoder (ModifiedWhisperEncoder, ultravox_model.py:803-994) */
  int32_t enc_layers, enc_d, enc_heads, enc_ffn, n_mels, enc_max_pos;
  int32_t enc_block; /* audio_latency_block_size (ultravox_model.py:834-863), 0 = no streaming mask */
  float ln_eps;      /* nn.LayerNorm eps (1e-5) */
  /* projector (UltravoxProjector, ultravox_model.py:745-800) */
  int32_t stack_factor, proj_hidden, proj_ln_mid;
  float proj_eps; /* RMSNorm eps (1e-6, ultravox_model.py:734) */
  /* language model (Llama family, reached at ultravox_model.py:328-334) */
  int32_t llm_layers, llm_d, llm_heads, llm_kv_heads, llm_head_dim, llm_inter, vocab;
  float rms_eps;
  /* backbone family behind AutoModelForCausalLM (ultravox_model.py:499-526).  UVX_LLM_LLAMA: RMSNorm w * round(x_hat), SwiGLU.
   * UVX_LLM_GEMMA (BASELINE config 5): GemmaRMSNorm x_hat * (1 + w) in f32, GeGLU (gelu_pytorch_tanh), inputs_embeds
   * multiplied by sqrt(hidden_size) in the model dtype INSIDE the model (transformers 4.51.3 GemmaModel.forward: text and
   * merged audio rows alike), head_dim independent of hidden_size / heads (256), lm_head tied to embed_tokens by the host. */
  int32_t llm_flavor;
  /* activation of the gated MLP ([3P] ACT2FN[text_config.hidden_act]): UVX_ACT_SILU (Llama), UVX_ACT_GELU_TANH
   * (gelu_pytorch_tanh, Gemma's default), UVX_ACT_GELU_ERF (exact GELU: Gemma checkpoints whose config says "gelu") */
  int32_t llm_act;
  /* Qwen3 (the reference's v0.6 recipe trains on Qwen/Qwen3-32B, ultravox/training/configs/v0.6_config_qwen3_32b.yaml): q_norm /
   * k_norm, an RMSNorm over head_dim on every head of q and k BEFORE the rotary embedding ([3P] modeling_qwen3.py Qwen3Attention).
   * Non-zero: uvx_llm_layer_t.q_norm / k_norm must be set, and training workspaces keep the un-normalised q | k rows per layer. */
  int32_t llm_qk_norm;
  /* Non-zero: uvx_llm_bwd* make the transposed weight copies they need ([K_in, N_out] of wqkv / wo / wgu / wd and of lm_head) ON
   * THE FLY, one layer ahead of the layer being differentiated, on an internal side stream into two alternating workspace
   * buffers - uvx_llm_layer_t.*_t and lm_head_t may then be NULL.  Halves the resident weight bytes (a 70B-parameter LLM then
   * trains on one 288 GB GPU: 141 GB of weights instead of 282) for one extra read + write of the weights per step. */
  int32_t llm_wt_stream;
  /* UVX_LLM_GEMMA3 (the reference's v0.6_config_gemma3_27b.yaml: the text stack of google/gemma-3-27b-it, [3P] modeling_gemma3.py):
   * llm_attn_scale = query_pre_attn_scalar ** -0.5 (0 = head_dim ** -0.5, every other family); llm_window = sliding_window of the
   * layers flagged in uvx_llm_weights_t.layer_local (Gemma-3: with their own rotary table; any other flavour - [3P] MistralConfig.sliding_window,
   * every layer flagged - with the one table).  Sequences of at most llm_window positions run those layers
   * as plain causal attention (incl. the fused backward); longer ones run the WINDOWED forms of the same kernels (a query sees the keys
   * in (q - window, q]: forward, dQ + dK/dV pair, chunked prefill), and the decode steps clamp the first visible cache slot. */
  float llm_attn_scale;
  int32_t llm_window;
  /* UltravoxProjector's activation (config.projector_act, ultravox_model.py:754-755): UVX_PROJ_SWIGLU (the default and every release config:
   * linear_1 -> SwiGLU halves the width -> linear_2 [D, hidden / 2]) or a plain activation of transformers' ACT2FN that keeps the width
   * (linear_2 [D, hidden], ln_mid over hidden): UVX_PROJ_SILU ("silu" / "swish"), UVX_PROJ_GELU_TANH ("gelu_pytorch_tanh"),
   * UVX_PROJ_GELU ("gelu", exact erf), UVX_PROJ_RELU. */
  int32_t proj_act;
} uvx_config_t;
#define UVX_PROJ_SWIGLU 0
#define UVX_PROJ_SILU 1
#define UVX_PROJ_GELU_TANH 2
#define UVX_PROJ_GELU 3
#define UVX_PROJ_RELU 4
#define UVX_ACT_SILU 0
#define UVX_ACT_GELU_TANH 1
#define UVX_ACT_GELU_ERF 2
#define UVX_LLM_LLAMA 0
#define UVX_LLM_GEMMA 1
/* Gemma-3 text stack: Gemma norms and GeGLU as UVX_LLM_GEMMA, plus a POST norm on each branch before its residual add
 * (uvx_llm_layer_t.ln1_post / ln2_post), q_norm / k_norm in the Gemma flavour (llm_qk_norm = 1), a second rotary table for the
 * sliding-window layers, llm_attn_scale; the sqrt(hidden) embedding scale is applied to the LOOKED-UP rows only (uvx_embed_merge:
 * [3P] Gemma3TextScaledWordEmbedding), never to inputs_embeds inside the model; head tied by the host. */
#define UVX_LLM_GEMMA3 2

/* Encoder weights.  Names follow the HF WhisperEncoder state dict; packing done once at
 * load time by the host:  wqkv = [q_proj*head_dim^-0.5 ; k_proj ; v_proj] ([3d, d]), bqkv likewise with a
 * zero k bias;  conv1_w [d, Kp1] with column k*n_mels + c = conv1.weight[:, c, k] zero padded to
 * Kp1 = roundup(3*n_mels, 64);  conv2_w [d, 3d] with column k*d + c = conv2.weight[:, c, k]. */
typedef struct {
  const void *ln1_w, *ln1_b, *wqkv, *bqkv, *wo, *bo, *ln2_w, *ln2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
  /* transposed copies ([K_in, N_out]) for the activation gradients of uvx_encoder_bwd; NULL for inference */
  const void *wqkv_t, *wo_t, *fc1_t, *fc2_t;
} uvx_enc_layer_t;
typedef struct {
  const void *conv1_w, *conv1_b, *conv2_w, *conv2_b, *pos;
  const uvx_enc_layer_t* layers; /* HOST array [enc_layers] of device-pointer records */
  const void *lnf_w, *lnf_b;
} uvx_encoder_weights_t;

/* LoRA on the attention projections of a tower (audio_model_lora_config / text_model_lora_config, ultravox_config.py:9-23;
 * applied by peft through apply_lora, ultravox_model.py:690-709): peft layouts, lora_A.weight [r, in] and lora_B.weight [out, r] in
 * the cfg dtype; result += lora_B(lora_A(x)) * scaling with scaling = lora_alpha / r.  Gradients: f32, same shapes.
 * q, k = q_proj / k_proj (the reference's default target_modules that exist in Whisper / Llama); v = v_proj; o = out_proj (Whisper) /
 * o_proj (the LLMs) - ABI 17; g, u, d = the MLP's linears (ABI 18): g = fc1 (Whisper) / gate_proj (the LLMs), u = up_proj (the LLMs only), d = fc2 /
 * down_proj.  A projection whose `a` is NULL is not adapted (its `b` and its gradient pointers are ignored). */
typedef struct { const void *a, *b; } uvx_lora_proj_t;
typedef struct { uvx_lora_proj_t q, k, v, o, g, u, d; } uvx_enc_lora_layer_t;
typedef struct {
  int32_t r;       /* 1..64 */
  float scaling;
  const uvx_enc_lora_layer_t* layers; /* HOST array [enc_layers] */
} uvx_encoder_lora_t;
typedef struct { float *a, *b; } uvx_lora_proj_grad_t;
typedef struct { uvx_lora_proj_grad_t q, k, v, o, g, u, d; } uvx_enc_lora_layer_grads_t;
typedef struct { const uvx_enc_lora_layer_grads_t* layers; } uvx_encoder_lora_grads_t;

/* multi_modal_projector.{ln_pre,linear_1,ln_mid|ln_post,linear_2}.weight (ultravox_model.py:749-766) */
typedef struct {
  const void *ln_pre, *w1, *ln_mid, *w2, *ln_post;
} uvx_projector_weights_t;
/* f32 gradient outputs, same shapes as the weights (ln_mid / ln_post: the one that exists) */
typedef struct {
  float *ln_pre, *w1, *ln_mid, *w2, *ln_post;
} uvx_projector_grads_t;

/* Llama layer: wqkv = [q;k;v] ([(H+2Hkv)*dh, D]), wgu = gate and up rows interleaved in 16-row blocks
 * ([2I, D]: rows 32k..32k+15 = gate rows 16k.., rows 32k+16..32k+31 = up rows 16k..); the *_t members are the
 * transposed copies ([K_in, N_out] -> nn.Linear layout of the transposed map) used for the frozen-weight
 * activation gradients; they may be NULL for forward-only use. */
typedef struct {
  const void *ln1, *wqkv, *wo, *ln2, *wgu, *wd;
  const void *wqkv_t, *wo_t, *wgu_t, *wd_t;
  /* family extras, NULL when the family has none.  bqkv: [q;k;v] projection biases [(H+2Hkv)*dh] (Qwen2: q_proj / k_proj / v_proj
   * carry a bias, o_proj does not - [3P] modeling_qwen2.py Qwen2Attention); q_norm, k_norm: [dh] (Qwen3, see llm_qk_norm). */
  const void *bqkv, *q_norm, *k_norm;
  /* Gemma-3: post_attention_layernorm / post_feedforward_layernorm [D] (ln2 is then pre_feedforward_layernorm) */
  const void *ln1_post, *ln2_post;
} uvx_llm_layer_t;
typedef struct {
  const void* embed;             /* [vocab, D] */
  const uvx_llm_layer_t* layers; /* HOST array [llm_layers] */
  const void* norm;              /* [D] */
  const void* lm_head;           /* [vocab, D] */
  const void* lm_head_t;         /* [D, vocab] or NULL */
  const float* rope_cos_sin;     /* [rope_len, head_dim/2, 2] f32 (cos, sin), built by the host */
  int32_t rope_len;
  /* Gemma-3 (else NULL): the rotary table of the sliding-window layers (rope_local_base_freq, same shape).  layer_local: a HOST array
   * [llm_layers] of flags, 1 = sliding-window ("local") layer of llm_window positions (Gemma-3's local layers; every layer of a Mistral
   * config with a sliding_window); NULL = no windowed layer */
  const float* rope_cos_sin_local;
  const int32_t* layer_local;
} uvx_llm_weights_t;

/* ============================== hot-path entry points ============================== */

/* K1 — log-mel frontend.  Replaces the [3P] WhisperFeatureExtractor call at
 * ultravox_processing.py:295-303.  pcm [B, L] f32 (L % 160 == 0, already padded as the reference pads);
 * out [B, n_mels, F_stride] f32, F = L/160 frames written.  Tables built by the host
 * (ultravox_amd/frontend.py): window[400], tw_cos/tw_sin [400][208], mel_fb [n_mels][208].
 * scratch: B * ceil(F/32) floats. */
int32_t uvx_logmel(void* stream, const float* pcm, const float* window, const float* tw_cos, const float* tw_sin,
                   const float* mel_fb, float* out, float* scratch, int32_t B, int32_t L, int32_t n_mels,
                   int32_t F_stride);

/* ModifiedWhisperEncoder.forward (ultravox_model.py:865-994), inference mode (frozen tower).
 * mel [B, n_mels, F] (f32 if mel_is_f32 else cfg dtype; cast like ultravox_model.py:383);
 * audio_lens [B] int64 mel-frame lengths or NULL (no padding mask); out [B, Te, d], Te = (F-1)/2+1. */
size_t uvx_encoder_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t F);
int32_t uvx_encoder_fwd(void* stream, const uvx_config_t* cfg, const uvx_encoder_weights_t* w, const void* mel,
                        int32_t mel_is_f32, const int64_t* audio_lens, int32_t B, int32_t F, void* out,
                        void* workspace, size_t ws_bytes);

/* UltravoxProjector.forward (ultravox_model.py:768-800) and its backward.  enc_out [B, Te, C];
 * out [B, Na, D] with Na = ceil(Te / stack_factor).  The forward leaves its activations in `workspace`;
 * the backward must be given the same (untouched) workspace.  dout [B, Na, D]; grads are f32 and
 * OVERWRITTEN. */
size_t uvx_projector_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t Te);
int32_t uvx_projector_fwd(void* stream, const uvx_config_t* cfg, const uvx_projector_weights_t* w, const void* enc_out,
                          int32_t B, int32_t Te, void* out, void* workspace, size_t ws_bytes);
int32_t uvx_projector_bwd(void* stream, const uvx_config_t* cfg, const uvx_projector_weights_t* w, const void* dout,
                          int32_t B, int32_t Te, const uvx_projector_grads_t* grads, void* d_enc_out, void* workspace,
                          size_t ws_bytes);
/* d_enc_out: NULL, or [B, Te, C] = d loss / d enc_out (needed only when the encoder itself trains: LoRA). */

/* Encoder under LoRA training (the reference's release configs train the projector plus rank-8
 * LoRA on the Whisper q_proj / k_proj).  uvx_encoder_fwd_train = uvx_encoder_fwd with the LoRA terms added and a
 * per-layer activation stash left in `workspace`; uvx_encoder_bwd turns d out [B, Te, d] into the LoRA gradients
 * (OVERWRITTEN, f32).  The layers need their transposed weight copies (uvx_enc_layer_t.*_t).  Same audio_lens for both. */
size_t uvx_encoder_train_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t F);
int32_t uvx_encoder_fwd_train(void* stream, const uvx_config_t* cfg, const uvx_encoder_weights_t* w,
                              const uvx_encoder_lora_t* lora, const void* mel, int32_t mel_is_f32, const int64_t* audio_lens,
                              int32_t B, int32_t F, void* out, void* workspace, size_t ws_bytes);
int32_t uvx_encoder_bwd(void* stream, const uvx_config_t* cfg, const uvx_encoder_weights_t* w, const uvx_encoder_lora_t* lora,
                        const void* d_out, const int64_t* audio_lens, int32_t B, int32_t F,
                        const uvx_encoder_lora_grads_t* grads, void* workspace, size_t ws_bytes);

/* Alt audio tower (BASELINE.json config 5): [3P] transformers Wav2Vec2Model.forward, the AutoModel branch of
 * UltravoxModel._create_audio_tower (ultravox_model.py:460-467, :476-485): the facebook/wav2vec2-large-960h family (GroupNorm after the first
 * conv layer, bias-free convs, post-LN encoder) and - round 5 - the layer-norm family (facebook/wav2vec2-large-lv60 / -960h-lv60-self:
 * LayerNorm after every conv layer, conv biases, pre-LN "stable" encoder).  Frozen tower: forward only.
 * input_values [B, L] (f32 if values_is_f32 else cfg dtype): the zero-mean / unit-variance waveform of the `input_values`
 * fallback (ultravox_processing.py:308); out [B, frames, d] with frames = uvx_wav2vec2_frames(cfg, L).  No attention mask
 * (group-norm wav2vec2 models are used without one).
 * Weight packing (host, once): conv0_w [C, 64] = conv_layers.0.conv.weight[:, 0, k] in column k, zero padded;
 * conv_w[i] (i >= 1) [C, k_i C] with column k C + c = conv_layers.i.conv.weight[:, c, k];  pos_w [G][d/G, K d/G] with
 * column k d/G + c = (weight-normed) pos_conv_embed.conv.weight[g d/G + o, c, k];  layers[] reuse uvx_enc_layer_t:
 * wqkv = [q_proj * head_dim^-0.5 ; k_proj ; v_proj], bqkv likewise, ln1 = layers.N.layer_norm, ln2 = final_layer_norm. */
typedef struct {
  int32_t dtype;
  int32_t n_conv, conv_dim;
  int32_t conv_kernel[8], conv_stride[8];
  int32_t d, heads, ffn, layers;
  int32_t pos_k, pos_groups;
  float ln_eps;
  /* round 5: the layer-norm ("-lv60") family next to the group-norm one.  feat_norm_layer = Wav2Vec2Config.feat_extract_norm == "layer": a
   * LayerNorm over the channels after EVERY conv layer (conv_ln_w / conv_ln_b) instead of the GroupNorm after the first; conv_bias: the conv
   * layers carry a bias (conv_b); stable_ln = do_stable_layer_norm: Wav2Vec2EncoderStableLayerNorm - pre-LN layers (layers.N.layer_norm in
   * front of the attention, final_layer_norm in front of the feed-forward) and encoder.layer_norm AFTER the last layer instead of before the first. */
  int32_t feat_norm_layer, conv_bias, stable_ln;
} uvx_w2v_config_t;
typedef struct {
  const void *conv0_w, *gn_w, *gn_b;
  const void* conv_w[8];
  const void *fp_ln_w, *fp_ln_b, *fp_w, *fp_b;
  const void *pos_w, *pos_b;
  const void *ln_w, *ln_b;
  const uvx_enc_layer_t* layers; /* HOST array [layers] */
  const void* conv_b[8];                       /* conv_layers.i.conv.bias [C] (conv_bias), else NULL */
  const void *conv_ln_w[8], *conv_ln_b[8];     /* conv_layers.i.layer_norm.{weight,bias} [C] (feat_norm_layer), else NULL */
} uvx_w2v_weights_t;
int32_t uvx_wav2vec2_frames(const uvx_w2v_config_t* cfg, int32_t L); /* [3P] _get_feat_extract_output_lengths; -1 if too short */
size_t uvx_wav2vec2_ws_bytes(const uvx_w2v_config_t* cfg, int32_t B, int32_t L);
int32_t uvx_wav2vec2_fwd(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const void* input_values,
                         int32_t values_is_f32, int32_t B, int32_t L, void* out, void* workspace, size_t ws_bytes);
/* ABI 17: the AutoModel tower under apply_lora (ultravox_model.py:460-467: `apply_lora(audio_tower, audio_model_lora_config)` wraps whatever tower was
 * loaded; :690-709) - adapters on the attention projections of every encoder layer (uvx_enc_lora_layer_t: q_proj / k_proj / v_proj / out_proj of
 * [3P] Wav2Vec2Attention), both encoder families (post-LN and do_stable_layer_norm).  uvx_wav2vec2_fwd_train = uvx_wav2vec2_fwd + the adapters'
 * terms + a per-layer stash in a workspace of uvx_wav2vec2_train_ws_bytes; uvx_wav2vec2_bwd walks it: d_out [B, frames, d] (the gradient
 * uvx_projector_bwd hands over) -> f32 gradients of every lora_A / lora_B.  The feature encoder, feature projection and positional conv carry no
 * adapter and receive no gradient (the walk stops at layer 0's input); uvx_enc_layer_t.*_t (transposed copies) are required. */
size_t uvx_wav2vec2_train_ws_bytes(const uvx_w2v_config_t* cfg, int32_t B, int32_t L);
int32_t uvx_wav2vec2_fwd_train(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const uvx_encoder_lora_t* lora,
                               const void* input_values, int32_t values_is_f32, int32_t B, int32_t L, void* out, void* workspace, size_t ws_bytes);
int32_t uvx_wav2vec2_bwd(void* stream, const uvx_w2v_config_t* cfg, const uvx_w2v_weights_t* w, const uvx_encoder_lora_t* lora, const void* d_out,
                         int32_t B, int32_t L, const uvx_encoder_lora_grads_t* grads, void* workspace, size_t ws_bytes);

/* embed_tokens + the in-place audio overwrite loop (ultravox_model.py:314-316, :390-394, :259-275).
 * input_ids [B, T] int64; audio_embeds [n_items, Na, D]; audio_batch_size [B] int64;
 * audio_token_start_idx [n_items] int64; audio_token_len [n_items] int32.  Later items overwrite earlier
 * ones exactly as the reference's sequential loop does.  scratch: (B*T + n_items) int32, kept for
 * uvx_merge_embeds_bwd (d audio_embeds = gather of d inputs_embeds rows; text rows get no gradient). */
int32_t uvx_embed_merge(void* stream, const uvx_config_t* cfg, const void* embed_table, const int64_t* input_ids,
                        const void* audio_embeds, const int64_t* audio_batch_size, const int64_t* audio_token_start_idx,
                        const int32_t* audio_token_len, int32_t B, int32_t T, int32_t n_items, int32_t Na,
                        void* inputs_embeds, int32_t* scratch);
int32_t uvx_merge_embeds_bwd(void* stream, const uvx_config_t* cfg, const void* d_inputs_embeds,
                             const int64_t* audio_token_start_idx, const int32_t* audio_token_len, int32_t B, int32_t T,
                             int32_t n_items, int32_t Na, void* d_audio_embeds, const int32_t* scratch);

/* LlamaForCausalLM.forward(inputs_embeds, attention_mask, labels) + ForCausalLMLoss
 * (reached at ultravox_model.py:328-334).  inputs_embeds [B, T, D]; attention_mask [B, T] int64 (1 = keep;
 * the kept range must be contiguous, as the collator produces) or NULL; labels [B, T] int64 or NULL;
 * logits [B, T, vocab] (cfg dtype) or NULL when only the loss is wanted and save_for_bwd = 0... it is
 * still computed internally; loss f32[1] or NULL.  With save_for_bwd != 0 the per-layer activations are
 * kept in `workspace` for uvx_llm_bwd. */
size_t uvx_llm_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t T, int32_t save_for_bwd);
int32_t uvx_llm_fwd(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                    const int64_t* attention_mask, const int64_t* labels, int32_t B, int32_t T, void* logits,
                    float* loss, int32_t save_for_bwd, void* workspace, size_t ws_bytes);
/* Activation-gradient backward of the frozen LLM (apply_lora r=0 freeze, ultravox_model.py:697-703):
 * d loss / d inputs_embeds [B, T, D], loss scaled by grad_scale (1 / gradient_accumulation_steps). */
int32_t uvx_llm_bwd(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels,
                    int32_t B, int32_t T, float grad_scale, void* d_inputs_embeds, void* workspace, size_t ws_bytes);
/* The adapter-training step's own pair: the same loss and gradients as uvx_llm_fwd(labels, logits = NULL, save_for_bwd = 1) +
 * uvx_llm_bwd, but the LAST layer's o_proj, MLP and final norm (and their gradients) are evaluated on the supervised rows
 * only - nothing downstream of the last attention mixes positions, and only positions whose next token is labelled enter
 * ForCausalLMLoss.  bf16 only; the two calls must be used together (the last layer's stash is row-compacted). */
int32_t uvx_llm_fwd_train(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                          const int64_t* attention_mask, const int64_t* labels, int32_t B, int32_t T, float* loss,
                          void* workspace, size_t ws_bytes);
int32_t uvx_llm_bwd_train(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels, int32_t B,
                          int32_t T, float grad_scale, void* d_inputs_embeds, void* workspace, size_t ws_bytes);
/* uvx_llm_bwd_train for a caller that needs no gradient below position `first_pos` of any sequence (ABI 19).  In the adapter-training step the
 * only consumer of d_inputs_embeds is the scatter back to the audio rows (_prepare_audio_embeds' inverse, uvx_merge_embeds_bwd), and under the
 * causal mask a position feeds later positions only: nothing trainable is reachable from the text before the first audio token, so
 * first_pos = min(audio_token_start_idx) over the batch.  Below the last layer the gradient tensors are then row-compacted to the positions
 * >= first_pos rounded down to a multiple of 16 (dgrad GEMMs, SwiGLU / RMSNorm backward on B * (T - first_pos) rows; the attention backward
 * still reads every key).  d_inputs_embeds rows at or above first_pos: bit-identical to uvx_llm_bwd_train's; rows below it (rounded down):
 * zeros.  Where the compacted form does not apply (tuning options 11 >= 2, 12 = 0 or 14 = 0) or first_pos < 16 the call IS uvx_llm_bwd_train. */
int32_t uvx_llm_bwd_train_from(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const int64_t* labels, int32_t B,
                               int32_t T, int32_t first_pos, float grad_scale, void* d_inputs_embeds, void* workspace, size_t ws_bytes);
/* labels == NULL: the saved logits already hold d loss / d logits (see uvx_llm_kl_loss). */

/* KL-distillation loss: UltravoxModel._compute_kl_loss (ultravox_model.py:200-256) with the
 * prediction / end-of-turn masks of _get_prediction_mask (:157-198) handed over as row pairs.  Call order:
 *   uvx_llm_fwd(teacher: alt embeddings, save_for_bwd = 0, logits -> teacher_logits [teacher_rows, vocab])
 *   uvx_llm_fwd(student: merged embeddings, save_for_bwd = 1, labels = NULL)
 *   uvx_llm_kl_loss(...)   -- reads the student logits kept in `workspace`, writes loss[0] and replaces them
 *                             by d loss / d logits (scaled by grad_scale)
 *   uvx_llm_bwd(labels = NULL, ...)
 * pair_row: int32 [2][B*T]: for student row r = b*T + t, slot 0 = teacher row paired with it among the prediction
 * positions, slot 1 = teacher row paired with it among the end-of-turn positions; -1 = none.  pair_w: f32 [2][B*T],
 * the weights (1 / n_pred and eot_loss_weight / n_eot: F.kl_div reduction="batchmean").
 * loss = sum_r sum_slot w * KL(softmax(teacher / temperature) || softmax(student / temperature)). */
int32_t uvx_llm_kl_loss(void* stream, const uvx_config_t* cfg, const void* teacher_logits, int64_t teacher_rows,
                        const int32_t* pair_row, const float* pair_w, int32_t B, int32_t T, float temperature,
                        float grad_scale, float* loss, void* workspace, size_t ws_bytes);

/* LLM under LoRA training (text_model_lora_config.r > 0): uvx_llm_fwd / uvx_llm_bwd with rank-r adapters on q_proj and
 * k_proj (peft layouts as for the encoder; q: [r, D] / [heads*head_dim, r], k: [r, D] / [kv_heads*head_dim, r]); the backward
 * also returns their f32 gradients.  uvx_llm_lora_t is the encoder's descriptor type. */
typedef uvx_encoder_lora_t uvx_llm_lora_t;
typedef uvx_encoder_lora_grads_t uvx_llm_lora_grads_t;
int32_t uvx_llm_fwd_lora(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const uvx_llm_lora_t* lora,
                         const void* inputs_embeds, const int64_t* attention_mask, const int64_t* labels, int32_t B, int32_t T,
                         void* logits, float* loss, int32_t save_for_bwd, void* workspace, size_t ws_bytes);
int32_t uvx_llm_bwd_lora(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const uvx_llm_lora_t* lora,
                         const int64_t* labels, int32_t B, int32_t T, float grad_scale, void* d_inputs_embeds,
                         const uvx_llm_lora_grads_t* grads, void* workspace, size_t ws_bytes);

/* The same with the LM head restricted to the rows that enter the loss (the prediction / end-of-turn positions: 256 of
 * 2528 at C2): rows = device list of positions b*T + t, ascending, host-known length.
 *   uvx_llm_fwd_rows(teacher, rows_t, n_t, logits_rows -> [n_t, vocab], save_for_bwd = 0)
 *   uvx_llm_fwd_rows(student, rows_s, n_s, NULL, save_for_bwd = 1)         -- compact logits + list stay in `workspace`
 *   uvx_llm_kl_loss_rows(teacher [n_t, vocab], pair [2][n_s] = index into the teacher's rows or -1, pair_w [2][n_s], ...)
 *   uvx_llm_bwd_rows(...)
 * bf16 only.  Nothing but the listed rows' logits leaves these calls, so the LAST layer's row-wise half (o_proj, MLP, final norm - and, in
 * uvx_llm_bwd_rows, their gradients) runs on the listed rows only (round 6; option 3 = 0 restores every row; Gemma-3: every row). */
int32_t uvx_llm_fwd_rows(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                         const int64_t* attention_mask, int32_t B, int32_t T, const int32_t* rows, int32_t n_rows,
                         void* logits_rows, int32_t save_for_bwd, void* workspace, size_t ws_bytes);
int32_t uvx_llm_kl_loss_rows(void* stream, const uvx_config_t* cfg, const void* teacher_logits_rows, const int32_t* pair,
                             const float* pair_w, int32_t B, int32_t T, int32_t n_rows, float temperature, float grad_scale,
                             float* loss, void* workspace, size_t ws_bytes);
int32_t uvx_llm_bwd_rows(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, int32_t B, int32_t T,
                         void* d_inputs_embeds, void* workspace, size_t ws_bytes);
/* ... from position first_pos on (the KL recipe's counterpart of uvx_llm_bwd_train_from, ABI 19: same conditions, same guarantees) */
int32_t uvx_llm_bwd_rows_from(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, int32_t B, int32_t T, int32_t first_pos,
                              void* d_inputs_embeds, void* workspace, size_t ws_bytes);

/* ---- inference: prefill + KV-cache decode.  Replaces the [3P] HF language_model.generate
 * that UltravoxModel.generate delegates to (ultravox_model.py:398-426) for GREEDY decoding.
 * kv_cache: [llm_layers][2 (k, v)][B][Tmax][kv_heads * head_dim] in the cfg dtype, caller-owned.
 * prefill: inputs_embeds [B, T, D], attention_mask [B, T] int64 or NULL (left padding allowed; position ids are
 * cumsum(mask) - 1 as HF derives them); fills cache rows [0, T), writes next_pos[b] = number of real tokens,
 * kv_start[b] = first real position, and the logits of the LAST position [B, vocab].
 * decode: one new token per sequence: token_embeds [B, D], positions[b] = next_pos[b] + step, cache row cur_len is
 * appended (cur_len = T + step), logits [B, vocab]. */
size_t uvx_kv_cache_bytes(const uvx_config_t* cfg, int32_t B, int32_t Tmax);
size_t uvx_llm_infer_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t T);
int32_t uvx_llm_prefill(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                        const int64_t* attention_mask, int32_t B, int32_t T, void* kv_cache, int32_t Tmax, int32_t* next_pos,
                        int32_t* kv_start, void* logits_last, void* workspace, size_t ws_bytes);
int32_t uvx_llm_decode(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* token_embeds,
                       const int32_t* positions, const int32_t* kv_start, void* kv_cache, int32_t Tmax, int32_t cur_len,
                       int32_t B, void* logits, void* workspace, size_t ws_bytes);
/* prefill of Tn FURTHER tokens per sequence on top of cur_len cached positions: what HF generate does when handed
 * past_key_values (infer.py:137-139, 208-213: conversation mode runs only input_ids[:, cache_len:]).  inputs_embeds
 * [B, Tn, D] (no padding inside the chunk); positions0[b] = RoPE position of the chunk's first token (next_pos[b] + tokens
 * decoded so far); cache rows [cur_len, cur_len + Tn) are appended; logits of the chunk's last position [B, vocab]. */
size_t uvx_llm_prefill_chunk_ws_bytes(const uvx_config_t* cfg, int32_t B, int32_t Tn, int32_t cur_len);
int32_t uvx_llm_prefill_chunk(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                              int32_t B, int32_t Tn, void* kv_cache, int32_t Tmax, int32_t cur_len, const int32_t* positions0,
                              const int32_t* kv_start, void* logits_last, void* workspace, size_t ws_bytes);
/* The same with the logits of EVERY new position, logits_all [B, Tn, vocab]: what the language model returns for
 * forward(past_key_values=...) without logits_to_keep (ultravox_model.py:328-334 -> [3P] LlamaForCausalLM.forward). */
int32_t uvx_llm_prefill_chunk_logits(void* stream, const uvx_config_t* cfg, const uvx_llm_weights_t* w, const void* inputs_embeds,
                                     int32_t B, int32_t Tn, void* kv_cache, int32_t Tmax, int32_t cur_len, const int32_t* positions0,
                                     const int32_t* kv_start, void* logits_all, void* workspace, size_t ws_bytes);
/* out[r] = argmax_v logits[r, v] (lowest index on ties, like torch.argmax) */
int32_t uvx_argmax(void* stream, int32_t dtype, const void* logits, int32_t rows, int32_t V, int64_t* out);
/* One step of the greedy generate() loop's bookkeeping on the device ([3P] GenerationMixin._sample with do_sample = False, what
 * ultravox_model.py:422-426 -> language_model.generate runs per token): for every row b
 *   tok = unfinished[b] ? argmax_v logits[b, v] : pad;   next_tokens[b] = sequences[b * stride + col] = tok;
 *   unfinished[b] &= tok not in eos_ids[0 .. n_eos);      positions[b] = positions0[b] + step (positions may be NULL);
 * counter[step & 1] (device, int32 x 2, both zero before step 0) receives the number of rows still unfinished - the host reads that one
 * word to decide whether to stop - and the other slot is cleared for the next step.  Calls of one loop must be ordered on one stream. */
int32_t uvx_greedy_select(void* stream, int32_t dtype, const void* logits, int32_t B, int32_t V, const int64_t* eos_ids, int32_t n_eos,
                          int64_t pad, int32_t* unfinished, int64_t* next_tokens, int64_t* sequences, int64_t stride, int64_t col,
                          const int32_t* positions0, int32_t* positions, int32_t step, int32_t* counter);

/* clip_grad_norm_(max_norm) + torch.optim.AdamW step over one flat parameter bucket (train.py:260,
 * config_base.py:149-154).  grad: f32 [n] (already DP-averaged).  state_dtype selects the storage of
 * param/m/v (bf16 mirrors the reference's bf16 optimizer state); with master != NULL the update runs on
 * f32 master weights and f32 moments.  scratch: 1025 floats. step counts from 1. */
int32_t uvx_adamw_clip_step(void* stream, int32_t state_dtype, void* param, float* master, const float* grad, void* m,
                            void* v, int64_t n, float max_norm, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t step, float* scratch);

/* ============================== data-parallel exchange ==============================
 * Replaces what torch DDP does under HF Trainer / accelerate (train.py:126-130, :282-288): the gradients of the
 * trainable parameters are summed over the ranks and divided by the world size, once per optimizer step, over the flat f32
 * bucket.  One communicator per process (one process per GPU); RCCL over xGMI, bound at run time (dlopen), so libuvx.so
 * does not link it and a host that never trains data-parallel does not need it.  rank 0 creates the 128-byte id and the
 * caller distributes it to the other ranks by its own channel (file, socket, MPI, torch.distributed's store). */
#define UVX_COMM_ID_BYTES 128
typedef struct uvx_comm uvx_comm_t;
int32_t uvx_comm_unique_id(uint8_t* id /* [UVX_COMM_ID_BYTES] */);
/* binds the communicator to the CURRENT HIP device of the calling thread; collective over all `world` ranks */
int32_t uvx_comm_init(uvx_comm_t** comm, int32_t rank, int32_t world, const uint8_t* id);
int32_t uvx_comm_world_size(const uvx_comm_t* comm);
int32_t uvx_comm_version(void);   /* RCCL's version code (e.g. 22606), 0 if RCCL cannot be loaded */
/* buf[i] = (sum over ranks of buf[i]) * scale, in place, asynchronous on `stream`; scale = 1 / world is DDP's gradient mean */
int32_t uvx_comm_allreduce_f32(uvx_comm_t* comm, void* stream, float* buf, int64_t n, float scale);
int32_t uvx_comm_destroy(uvx_comm_t* comm);

/* ============================== single-op entry points (tests, probes) ============================== */

typedef struct {
  const void* A; const void* B; void* C; const void* bias; const void* residual;
  int32_t M, N, K, lda, ldb, ldc, ldr;
  int32_t res_mod, batch;
  int64_t stride_a, stride_b, stride_c, stride_r;
  int32_t act;        /* 0 none, 1 exact-erf GELU; bf16 only: 2 = GELU that keeps its pre-activation - C = acc * alpha + bias (rounded), C2 [M, N] (row
                       * stride ldc2) = gelu(C); 3 = GELU backward - C = (acc * alpha, rounded) * gelu'(C2), C2 [M, N] = the saved pre-activation.
                       * 2 and 3 equal uvx_gemm + uvx_gelu / uvx_gelu_bwd bit for bit; no residual, epilogue, f32 output or batch with them */
  int32_t out_f32;    /* bf16 inputs, f32 output (weight gradients) */
  int32_t accumulate; /* C += (f32 output only) */
  float alpha;
  /* fused LlamaMLP epilogues (bf16 only; weights/activations in the interleaved gate|up layout of weights.py: 16-column
   * gate block, then the matching 16-column up block).  epilogue 1: C [M, N] = gate|up pre-activations,
   * C2 [M, N/2] = silu(gate) * up.  epilogue 2: the tile is d act [M, N]; with C2 = gate|up [M, 2N] (input) it is turned
   * into d gate|up written to C [M, 2N] (ldc >= 2N). */
  void* C2; int32_t ldc2; int32_t epilogue;
  int32_t b_kn;       /* ABI 16, bf16 only: 1 = B is stored [K, N] (row stride ldb) - C = A . B, the form a dgrad d x = d y . W takes on the forward weight
                       * W [N_out, N_in] as it lies (no transposed copy); N % 8 == 0; bit-identical to uvx_gemm on the transposed matrix */
  int32_t reserved_;
} uvx_gemm_desc_t;
/* C[M,N] = act(alpha * A[M,K] . B[N,K]^T + bias[N]) + residual — torch.nn.Linear semantics. */
int32_t uvx_gemm(void* stream, int32_t dtype, const uvx_gemm_desc_t* desc);
/* uvx_gemm for problems of a few hundred rows (round 5: the prefill of generate() - one or two prompts of 30 s of audio + text are
 * 316 / 632 rows, [3P] language_model.generate's first forward, ultravox_model.py:398-426): with 160- / 256-row tiles such a problem has
 * fewer tiles than the chip has CUs, so the K loop of every tile is cut over `s` blocks (tiles x s ~ one round of the 256 CUs) that
 * write f32 partial tiles into `workspace`, and one more kernel sums them in a fixed order (bit-reproducible) and applies the epilogue
 * with uvx_gemm's arithmetic and rounding points.  Results agree with uvx_gemm to f32 summation order.  bf16, batch 1, N % 8 == 0 and
 * 16-byte-aligned operands; anything else (and every problem the cost model would not split) runs exactly as uvx_gemm.
 * force_split: 0 = the cost model decides, 1 = never split, s > 1 = split by s (tests / probes).  workspace: uvx_gemm_splitk_ws_bytes(M, N)
 * bytes (at most 128 MiB), caller-owned, free for reuse once the call's work has run.  uvx_llm_prefill* take theirs from their workspace. */
size_t uvx_gemm_splitk_ws_bytes(int32_t M, int32_t N);
int32_t uvx_gemm_splitk(void* stream, int32_t dtype, const uvx_gemm_desc_t* desc, void* workspace, size_t ws_bytes, int32_t force_split);
/* host only (no GPU work): the split factor (1 = none) the cost model picks for a bf16 problem with `ws_bytes` of scratch; *variant
 * (may be NULL) receives the tile variant */
int32_t uvx_gemm_pick_split(int32_t M, int32_t N, int32_t K, size_t ws_bytes, int32_t* variant);
/* probes/tests: force the bf16 GEMM tile variant (-1 auto, 0 = 128x128, 1..4 = {128,160,192,256} x 256) */
int32_t uvx_gemm_force_variant(int32_t variant);
/* C = epilogue(RMSNorm(A; norm_w, eps) . B^T) - LlamaRMSNorm (flavor 0) / GemmaRMSNorm (1) feeding an nn.Linear, the pair the
 * decode step runs twice per layer ([3P] LlamaDecoderLayer: input_layernorm -> q|k|v, post_attention_layernorm -> gate|up).  bf16 with
 * at most 2 rows (and K <= 16384) - or, with option 24 = 1, 3..16 rows with K % 2048 == 0 -: ONE launch, the norm applied while the activation rows are
 * staged (resp. to the MFMA activation fragments in registers); anything else: the two launches
 * it replaces (norm_out [M, K] is the scratch for that case, may be NULL when the fused kernel is known to apply).  lda must equal K. */
int32_t uvx_gemm_rmsnorm(void* stream, int32_t dtype, const uvx_gemm_desc_t* g, const void* norm_w, float eps, int32_t flavor,
                         void* norm_out);
/* probes (same-box A/B inside bench.py): key 1 = 16-byte epilogue loads/stores (default 1), key 2 = SwiGLU backward fused
 * into the down-projection dgrad GEMM (0 = separate kernel, 1 = round 2's fragment-layout epilogue (measured neutral),
 * 2 = whole-line epilogue through the LDS stage (round 3)), key 3 = LM head / CE / head dgrad on the supervised
 * rows only (default 1; must not change between uvx_llm_fwd and uvx_llm_bwd), key 4 = weight-streaming GEMM kernels for
 * few-row problems (the decode step; default 1: row-streaming kernel for M <= 2, MFMA kernel with the weights staged through LDS for
 * M = 3..64 when K % 2048 == 0, straight fragment loads for other M <= 16; 2 = straight fragment loads for every M <= 16 - round 2's
 * kernel - and the tiled kernels above, for A/B; 0 = the tiled kernels), key 5 = the streamed weight transposes (llm_wt_stream) use plain instead of
 * non-temporal loads / stores (default 0), key 11 = number of LLM layer chains: the batch
 * is cut into that many slices whose layer chains run on as many streams (default 1 = one chain on the caller's stream, at most 4; which of 1 / 2 is
 * faster depends on the box: UltravoxTrainer.autotune_schedule times both; uvx_llm_fwd* / uvx_llm_bwd*: same kernels on the same rows, bit-identical results; the side streams are
 * forked from and joined into the caller's stream by events, so the call stays stream-ordered for the caller), key 12 = bf16
 * attention kernels read V^T / Q^T / K^T / dO^T out of the natural tiles with the transposing LDS read instead of from
 * transposed copies in global memory (default 1; 0 restores the copies: heads_transpose + the *_t staging), key 13 = the fused attention backward
 * kernel for head_dim 128 / causal / at most 320 positions (default 1; 0 = the dQ + dK/dV kernel pair), key 14 = the LLM's attention backward writes dq / dk
 * RoPE-inverted from its own epilogues instead of a separate pass over d_qkv (default 1; bit-identical), key 15 = TIMING
 * PROBE ONLY (default 0): bit mask of kernel classes that are not launched (results are garbage; what the class costs inside
 * the overlapped schedule): 1 LLM attention backward, 2 LLM attention forward, 4 SwiGLU backward, 8 RMSNorm backward,
 * 16 RMSNorm forward, 32 RoPE forward, 64 encoder attention, 128 encoder LayerNorm, key 16 = the prefill's rotary embedding and KV-cache
 * append in one launch per layer (default 1; 0 = the rope + append pair; bit-identical), key 17 = a split-K linear of generate() that is
 * followed by an RMSNorm (o_proj -> post_attention_layernorm, down_proj -> the next layer's input_layernorm) has that norm computed by its
 * reduce kernel (default 1; 0 = the separate rmsnorm launch; bit-identical), key 18 = 1: the RMSNorm forward of plain rows runs the two-pass kernel
 * instead of the one that keeps the row in registers (default 0; bit-identical; A/B), key 19 = tile form of the head_dim-64 attention backward pair (the
 * Whisper tower under LoRA training): 0 = two 16-query tiles per wave in the dQ kernel (default), 1 = one tile per wave in both kernels (rounds 1-5), 2 = two in
 * both, 3 = two in the dK/dV kernel only, 4 = 64-row steps, 5 / 6 = eight-wave blocks (all bit-identical; A/B), key 20 = 1: the head_dim-64 forward kernel takes
 * its row max through ds_bpermute shuffles instead of v_permlane swaps (default 0; bit-identical; A/B), key 21 = 1: the training tower's GELU and GELU backward
 * run as separate kernels instead of in the fc1 / fc2-dgrad GEMM epilogues (uvx_gemm_desc_t.act 2 / 3; default 0; bit-identical; A/B), key 23 = 1: the decode step's rotary embedding and
 * KV-cache append of the new token run as their own launch per layer instead of inside the grouped decode-attention kernel (default 0; bit-identical; A/B)., key 22 = 1: the training tower's q_proj / k_proj adapter products run as
 * separate launches instead of paired ones (lora_down2 / lora_up2 / one partial-sum launch for the four weight gradients; default 0; bit-identical; A/B),
 * key 24 = 1: uvx_gemm_rmsnorm / the decode step at 3..16 rows (K % 2048 == 0) normalise the activation fragments inside the staged weight-streaming
 * kernel instead of running rmsnorm + linear as two launches (default 0: measured 19-32 % slower per token; same arithmetic and rounding points, rstd may
 * differ in the last bit; A/B), key 25: the head_dim-128 causal attention forward under grouped-query attention (Hq / Hkv a multiple of 4, T <= 512) - 0 (default):
 * blocks whose four waves take four query heads of one KV head when the launch has >= 512 of them, 5: never, 1 / 3 / 4: always, with two / three / one
 * query tiles per wave (bit-identical in every form; A/B), key 26: weight rows in flight per wave of the one-row decode GEMV - 0 (default): 2, 4 / 8 / 16: that many
 * (agree to rounding, not bitwise; A/B). */
int32_t uvx_set_option(int32_t key, int32_t value);
/* the current value of a tuning option (-1: unknown key) */
int32_t uvx_get_option(int32_t key);
/* host only (no GPU work): the bf16 GEMM tile variant the cost model picks for this problem */
int32_t uvx_gemm_pick_variant(int32_t M, int32_t N, int32_t K, int32_t batch);
/* probes: use `variant` for problems of exactly this shape (in-situ A/B inside bench.py); variant < 0 clears the table */
int32_t uvx_gemm_override_variant(int32_t M, int32_t N, int32_t K, int32_t variant);
/* probe of the gfx950 transposing LDS read (ds_read_b64_tr_b16) that the bf16 attention kernels rely on for their transposed
 * operands: one wave; lane l supplies the BYTE offset addr[l] (8-byte aligned, < 8184) into an LDS image whose 16-bit element e
 * holds the value e; out[4 l + j] = element j the instruction returned to lane l.  tests/test_kernels_gpu.py pins the semantics. */
int32_t uvx_probe_lds_tr(void* stream, const int32_t* addr, int32_t* out);
/* probes: force the attention forward q-tile count per wave (1 or 2; 0 = automatic) */
int32_t uvx_attention_force_qt(int32_t qt);
int32_t uvx_layernorm(void* stream, int32_t dtype, const void* x, const void* w, const void* b, void* y, int32_t rows,
                      int32_t cols, float eps);
int32_t uvx_rmsnorm(void* stream, int32_t dtype, const void* x, const void* w, void* y, int32_t rows, int32_t cols,
                    float eps);
int32_t uvx_rmsnorm_bwd(void* stream, int32_t dtype, const void* dy, const void* x, const void* w, const void* dx_add,
                        void* dx, float* dw, int32_t rows, int32_t cols, float eps);
int32_t uvx_swiglu(void* stream, int32_t dtype, const void* in, void* out, int32_t rows, int32_t half, int32_t gate_first);
int32_t uvx_swiglu_bwd(void* stream, int32_t dtype, const void* dout, const void* in, void* din, int32_t rows,
                       int32_t half, int32_t gate_first);
int32_t uvx_rope(void* stream, int32_t dtype, void* x, const float* cos_sin, int32_t rows, int32_t T, int32_t n_heads,
                 int32_t head_dim, int32_t ld, int32_t inverse);
/* Qwen3 q_norm / k_norm + rotary embedding in one pass, in place on the q | k columns of qkv [rows, ld] (heads of width head_dim:
 * Hq query heads, then Hkv key heads; position of row r = r % T); wq / wk [head_dim]; raw = NULL or [rows, (Hq + Hkv) * head_dim],
 * receives the un-normalised q | k rows.  uvx_qk_norm_bwd: d_qkv's q | k columns hold the gradient of the normalised (pre-rotary)
 * rows on entry and the gradient of the raw rows on return. */
/* flavor: 0 = w * round(x_hat) (Qwen3RMSNorm == LlamaRMSNorm), 1 = x_hat * (1 + w) in f32 with one rounding (Gemma3RMSNorm) */
int32_t uvx_qk_norm_rope(void* stream, int32_t dtype, void* qkv, const void* wq, const void* wk, void* raw, const float* cos_sin,
                         int32_t rows, int32_t T, int32_t Hq, int32_t Hkv, int32_t head_dim, int32_t ld, float eps, int32_t flavor);
int32_t uvx_qk_norm_bwd(void* stream, int32_t dtype, void* d_qkv, const void* raw, const void* wq, const void* wk, int32_t rows,
                        int32_t Hq, int32_t Hkv, int32_t head_dim, int32_t ld, float eps, int32_t flavor);

typedef struct {
  const void *q, *k, *v; /* [B, T, H, D] views with token strides ldq/ldk/ldv (elements) */
  void* o;               /* [B, T, Hq*D], token stride ldo */
  float* lse;            /* [B, Hq, T] log2-domain log-sum-exp (needed for backward) or NULL */
  const int32_t *kv_start, *kv_len; /* [B] valid key range or NULL */
  int32_t B, T, Hq, Hkv, D, ldq, ldk, ldv, ldo, causal, block;
  float scale;
  /* backward only */
  const void* dout; void *dq, *dk, *dv; int32_t lddq, lddk, lddv;
  int32_t window; /* > 0 with causal: sliding window, a query sees keys in (q - window, q] only; 0 = none */
} uvx_attn_desc_t;
size_t uvx_attention_ws_bytes(int32_t dtype, const uvx_attn_desc_t* d, int32_t backward);
int32_t uvx_attention_fwd(void* stream, int32_t dtype, const uvx_attn_desc_t* d, void* workspace, size_t ws_bytes);
int32_t uvx_attention_bwd(void* stream, int32_t dtype, const uvx_attn_desc_t* d, void* workspace, size_t ws_bytes);

/* encoder-LoRA backward pieces: LayerNorm backward for a frozen affine (dx [+ dx_add] only), GELU as a separate pass
 * (the arithmetic of the fused GEMM epilogue) and its exact-derivative backward; n = element count (multiple of 8) */
int32_t uvx_layernorm_bwd(void* stream, int32_t dtype, const void* dy, const void* x, const void* w, const void* dx_add,
                          void* dx, int32_t rows, int32_t cols, float eps);
int32_t uvx_gelu(void* stream, int32_t dtype, const void* pre, void* out, int64_t n);
int32_t uvx_gelu_bwd(void* stream, int32_t dtype, const void* dout, const void* pre, void* din, int64_t n);

/* scratch: 2 + B*T floats */
int32_t uvx_ce_loss(void* stream, int32_t dtype, const void* logits, const int64_t* labels, float* loss, void* dlogits,
                    int32_t B, int32_t T, int32_t V, int32_t ld, float grad_scale, float* scratch);

/* the kernel behind uvx_llm_kl_loss on caller-owned logits [rows, V]; dlogits may alias student_logits or be NULL;
 * scratch: rows floats */
int32_t uvx_kl_loss(void* stream, int32_t dtype, const void* student_logits, const void* teacher_logits,
                    const int32_t* pair_row, const float* pair_w, float* loss, void* dlogits, int64_t rows, int32_t V,
                    int32_t ld_student, int32_t ld_teacher, float temperature, float grad_scale, float* scratch);

/* ---- live kernel timing (bench.py roofline leg): HIP events on the launch stream around every GEMM.
 * uvx_prof_end fills out[class*4 + {0: launches, 1: total ms, 2: algorithmic FLOPs, 3: algorithmic bytes}]
 * for class 0 = bf16 MFMA GEMM (classes 1.. reserved) and synchronises on the recorded events. */
int32_t uvx_prof_begin(void);
int32_t uvx_prof_enable(int32_t on); /* pause (0) / resume (1) the recording inside a region; the records so far are kept */
int32_t uvx_prof_end(double* out, int32_t n_classes);
/* wall time (ms) during which at least one launch of the class was executing: the union of the event intervals (= the summed
 * durations on one stream; with the two-stream LLM schedule, uvx_set_option(11, 2), launches of the two chains overlap).
 * Call before uvx_prof_end; negative on error. */
double uvx_prof_union_ms(int32_t cls);
/* per-launch GEMM records of the region (call before uvx_prof_end): out[i*6 + {M,N,K,batch,variant,ms}] */
int32_t uvx_prof_records(double* out, int32_t max_records);

#ifdef __cplusplus
}
#endif
#endif /* UVX_H_ */
