"""Static resource table of every kernel in libuvx (no GPU needed): registers, scratch (spills), LDS and the occupancy the
compiler derives, from hipcc's -Rpass-analysis=kernel-resource-usage remarks, with the package's own build flags.
usage: python tools/isa_resources.py > profiles/rNN_isa_resources.txt"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from ultravox_amd import build as B  # noqa: E402

rows = []
for src in sorted(B.CSRC.glob("*.hip")):
    cmd = [B._hipcc(), *B.FLAGS, *B.EXTRA_FLAGS.get(src.name, []), "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only",
           "-c", str(src), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)", line)
        if not m:
            continue
        k, v = m.groups()
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip() or v
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*$", "", name).replace("unsigned short", "bf16").replace("void ", "")
            cur = {"file": src.name, "kernel": name}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = int(v)
print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the package's build flags; scratch > 0 would mean register spills")
print(f"{'file':<18}{'kernel':<62}{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'scratch':>8}{'LDS B':>8}{'occ':>4}")
for r in rows:
    print(f"{r['file']:<18}{r['kernel'][:61]:<62}{r.get('VGPRs', 0):>5}{r.get('AGPRs', 0):>5}{r.get('TotalSGPRs', 0):>5}"
          f"{r.get('ScratchSize', 0):>8}{r.get('LDS', 0):>8}{r.get('Occupancy', 0):>4}")
spills = [r for r in rows if r.get("ScratchSize", 0) > 0]
print(f"# {len(rows)} kernels, {len(spills)} with scratch: {[r['kernel'] for r in spills]}")
