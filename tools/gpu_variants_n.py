"""Like gpu_variants.py at another size:  N=512 python tools/gpu_variants_n.py lib1.so lib2.so ..."""
import os, subprocess, sys, re
N = os.environ.get("N", "512"); POL = os.environ.get("POL", "P"); DT = os.environ.get("DT", "f32")
best = {}
for r in range(int(os.environ.get("ROUNDS", "2"))):
    for lib in sys.argv[1:]:
        env = dict(os.environ, CMBL_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, "tools/gpu_time.py", N, POL, DT], env=env, capture_output=True, text=True).stdout
        nums = {m[0].strip(): float(m[1]) for m in re.findall(r"^(\S+)\s+([0-9.]+) ms", out, flags=re.M)}
        best[lib] = nums if lib not in best else {k: min(v, best[lib].get(k, v)) for k, v in nums.items()}
for lib, n in best.items():
    print("MIN", os.path.basename(lib), " ".join(f"{k} {v:.3f}" for k, v in n.items()), flush=True)
