"""bench.py's host-side definitions, checked without a GPU: the algorithmic FLOP count against SURVEY.md §8(d)'s figures
for C2 / C3 and the lookup of the committed PMC traffic summary."""
import json
import os

import pytest

import bench
from ultravox_amd.config import UltravoxConfig


def test_flops_per_sample_matches_the_survey_figures():
    c2 = UltravoxConfig(audio_model_id="openai/whisper-medium", text_model_id="meta-llama/Meta-Llama-3-8B-Instruct",
                        hidden_size=4096, stack_factor=8, projector_ln_mid=True)
    f = bench.flops_per_sample(c2, 30.0, n_text=128, n_supervised=32)
    assert f["encoder"] / 1e9 == pytest.approx(1138.07, abs=0.01)          # E
    assert f["projector"] / 1e9 == pytest.approx(15.77, abs=0.01)          # P
    assert f["step_full_head"] / 1e9 == pytest.approx(10749.8, abs=0.2)    # 10 749.8 GFLOP per sample with the all-rows head
    assert f["step"] < f["step_full_head"]                                 # the quoted MFU counts the supervised rows only
    c3 = UltravoxConfig(audio_model_id="openai/whisper-large-v3", text_model_id="meta-llama/Meta-Llama-3-8B-Instruct",
                        hidden_size=4096, stack_factor=8, projector_ln_mid=True)
    g = bench.flops_per_sample(c3, 30.0)
    assert g["encoder"] / 1e9 == pytest.approx(2273.77, abs=0.01) and g["projector"] / 1e9 == pytest.approx(18.92, abs=0.01)
    assert g["step_full_head"] / 1e9 == pytest.approx(11894.9, abs=0.2)


def test_pmc_traffic_lookup_takes_the_newest_summary_and_never_raises(tmp_path):
    assert bench.pmc_traffic_per_launch(str(tmp_path)) is None and bench.pmc_traffic_per_launch("/nonexistent") is None
    for name, v in (("r01_pmc_traffic.json", 1.0), ("r02_pmc_traffic.json", 2.0)):
        json.dump({"traffic_bytes_per_launch": v}, open(tmp_path / name, "w"))
    assert bench.pmc_traffic_per_launch(str(tmp_path)) == 2.0
    (tmp_path / "r03_pmc_traffic.json").write_text("{broken")
    assert bench.pmc_traffic_per_launch(str(tmp_path)) is None
    committed = bench.pmc_traffic_per_launch()
    assert committed is None or committed > 1e8                            # bytes per GEMM launch at C2


def test_gpus_n_without_torchrun_becomes_the_launcher(monkeypatch):
    """`python bench.py --gpus 4` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run
    (one rank per GPU, 127.0.0.1 rendezvous) and forwards the exit code - it must not exit(2) asking for torchrun."""
    import subprocess
    import sys
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_ranks_synthesise_distinct_shards():
    """bench.py --gpus N: every rank builds its OWN shard (seeded by its rank) - a weak-scaling run whose ranks all train on
    the same clips would still print a plausible line."""
    import torch
    from ultravox_amd.synthetic import synthetic_batch
    cfg = UltravoxConfig(audio_config=dict(d_model=128, encoder_layers=1, encoder_attention_heads=2, encoder_ffn_dim=256),
                         text_config=dict(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
                                          num_key_value_heads=2, vocab_size=512, eos_token_id=2), hidden_size=256)
    shards = [synthetic_batch(cfg, 2, 1.0, n_text=16, audio_start=4, n_supervised=4, rank=r) for r in range(4)]
    for i in range(4):
        for j in range(i + 1, 4):
            assert not torch.equal(shards[i]["pcm"], shards[j]["pcm"]) and not torch.equal(shards[i]["input_ids"], shards[j]["input_ids"])
    again = synthetic_batch(cfg, 2, 1.0, n_text=16, audio_start=4, n_supervised=4, rank=2)
    assert torch.equal(again["pcm"], shards[2]["pcm"]) and torch.equal(again["input_ids"], shards[2]["input_ids"])
    assert all(s["input_ids"].shape == shards[0]["input_ids"].shape for s in shards)        # weak scaling: same shapes per rank


def test_decode_roofline_numerator_is_the_llm_weight_stream():
    """`bench.py --workload c4`: bytes one decode step must stream = every layer's q|k|v, o, gate|up, down + the LM head in bf16
    (Llama-3.3-70B: 69.5 B parameters -> 139 GB; at 8 TB/s a 17.4 ms floor per token)."""
    import bench
    from ultravox_amd.config import UltravoxConfig
    wl = bench.WORKLOADS["c4"]
    assert wl["inference"] and wl["B"] == 1 and wl["new_tokens"] == 32
    cfg = UltravoxConfig(audio_model_id=wl["audio"], text_model_id=wl["text"], hidden_size=4096, stack_factor=8, projector_ln_mid=True)
    b = bench.llm_weight_bytes(cfg)
    assert abs(b - 2 * (80 * ((64 + 16) * 128 * 8192 + 8192 * 8192 + 3 * 8192 * 28672) + 128256 * 8192)) < 1
    assert 138e9 < b < 140e9


def test_decode_traffic_lookup_reads_the_committed_pmc_summary(tmp_path):
    """`bench.py --workload c4` reports roofline.traffic from the newest profiles/rNN_pmc_decode_traffic.json and never raises."""
    import json
    import bench
    assert bench.pmc_decode_traffic_per_token(str(tmp_path)) is None
    (tmp_path / "r03_pmc_decode_traffic.json").write_text(json.dumps({"traffic_bytes_per_token": 1.0}))
    (tmp_path / "r04_pmc_decode_traffic.json").write_text(json.dumps({"traffic_bytes_per_token": 2.5}))
    assert bench.pmc_decode_traffic_per_token(str(tmp_path)) == 2.5
    (tmp_path / "r05_pmc_decode_traffic.json").write_text("not json")
    assert bench.pmc_decode_traffic_per_token(str(tmp_path)) is None
    got = bench.pmc_decode_traffic_per_token()          # the committed record: within 1 % of the weight bytes of Llama-3.3-70B
    assert got is not None and abs(got / 139003428864.0 - 1.0) < 0.01


def test_parity_object_cites_the_newest_committed_full_depth_record(tmp_path):
    """bench.py's `parity` object (say in the line what the bf16 path's distance to the oracle is instead of leaving
    north_star's 1e-3 implicit): read from profiles/rNN_parity/<workload>_full_depth.json, newest round first, never raises."""
    import bench
    assert bench.parity_record("c2", str(tmp_path)) is None and bench.parity_record("c2", "/nonexistent") is None
    for rnd, v in (("r03", 0.5), ("r11", 0.25)):
        (tmp_path / f"{rnd}_parity").mkdir()
        (tmp_path / f"{rnd}_parity" / "c2_full_depth.json").write_text(json.dumps(
            {"stages": {"logits": {"rel_l2": v, "max_abs": 1.0, "ref_rms": 1.0}}, "grads_rel_l2": {"a": 0.1, "b": 0.2}}))
    (tmp_path / "r12_parity").mkdir()          # a newer round without this workload's record: the older record still answers
    got = bench.parity_record("c2", str(tmp_path))
    assert got["logits_rel_l2_vs_f32_oracle"] == 0.25 and got["projector_grads_rel_l2_vs_f32_oracle_max"] == 0.2 and "r11_parity" in got["source"]
    (tmp_path / "r12_parity" / "c2_full_depth.json").write_text("{not json")
    assert bench.parity_record("c2", str(tmp_path)) is None
    committed = bench.parity_record("c2")
    assert committed is not None and 1e-3 < committed["logits_rel_l2_vs_f32_oracle"] < 2.8e-2      # the bar of tests/test_c2_full_depth_gpu.py


def test_live_traffic_measurement_parses_the_counter_csv_and_never_raises(tmp_path, monkeypatch):
    """bench.py measures roofline.traffic in the run itself (two rocprofv3 --pmc sub-runs); the host-side pieces: the per-launch mean over the
    gemm_nt rows of a counter_collection.csv, and the fall-back to the committed summary when the tool is missing or a sub-run fails."""
    import bench
    csv_file = tmp_path / "p_counter_collection.csv"
    csv_file.write_text("Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n"
                        '1,"void (anonymous namespace)::gemm_nt_bf16_ph8_kernel<256, 2, 0, false, 2, false>(GemmArgs)",FETCH_SIZE,300.0\n'
                        '2,"void (anonymous namespace)::rmsnorm_fwd_k<unsigned short>(...)",FETCH_SIZE,9999.0\n'
                        '3,"void (anonymous namespace)::gemm_nt_bf16_kernel(GemmArgs)",FETCH_SIZE,100.0\n')
    assert bench.pmc_gemm_counter_per_launch(str(csv_file)) == (2, 200.0)
    import shutil
    monkeypatch.setattr(shutil, "which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda p: False)
    got, why = bench.measure_traffic_live(["--workload", "c2"])
    assert got is None and "rocprofv3" in why
    monkeypatch.undo()
    # a tool that "runs" but leaves no CSV behind: reported, not raised
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nexit 0\n")
    fake.chmod(0o755)
    monkeypatch.setattr(shutil, "which", lambda name: str(fake))
    got, why = bench.measure_traffic_live(["--workload", "c2"], timeout_s=20)
    assert got is None and "sub-run failed" in why
