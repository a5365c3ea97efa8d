"""Per-kernel register / spill / LDS table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kres.py linear_bf16_tile.hip [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
       *sys.argv[2:], "-Rpass-analysis=kernel-resource-usage", "-c", sys.argv[1], "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, cwd=os.path.join(ROOT, "pytorch-models_amd", "csrc"), capture_output=True, text=True).stderr
rows, cur = [], None
for l in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", l)
    if not m:
        if "error" in l:
            print(l)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        n = t.split(":", 1)[1].strip()
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("void (anonymous namespace)::", "").split("(")[0]
        cur = {"name": n}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print(f"{r['name'][-64:]:64s} VGPR {r.get('VGPRs')} AGPR {r.get('AGPRs')} spillV {r.get('VGPRs Spill')} spillS {r.get('SGPRs Spill')} "
          f"scratch {r.get('ScratchSize [bytes/lane]')} occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}")
