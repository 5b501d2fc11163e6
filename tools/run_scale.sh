#!/bin/bash
# Turnkey scaling run on one multi-GPU node (not launched by the builder: the boxes gpurun hands out have one GPU).
#   bash tools/run_scale.sh [out_dir] [steps] [warmup]
# Runs bench.py at N = 1, 2, 4, 8 (capped at the number of visible GPUs), one rank per GPU over RCCL (`nccl` backend), and keeps one
# JSON line per N in <out_dir>/scale_N<N>.json plus scale_summary.txt.  For N > 1 the line carries "collective": backend, world size as
# the collective library reports it, and the device / PCI bus id / UUID of every rank -- the evidence that N distinct GPUs took part.
# Weak scaling: every rank evaluates its own chain (different seeds), value = N * steps / max-over-ranks time.
set -u
cd "$(dirname "$0")/.." || exit 1
out=${1:-gpurun_out/scale}; steps=${2:-200}; warm=${3:-5}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $ngpu" | tee "$out/scale_summary.txt"
for n in 1 2 4 8; do
  [ "$n" -gt "$ngpu" ] && { echo "N=$n skipped (only $ngpu GPUs)" | tee -a "$out/scale_summary.txt"; continue; }
  # plain calls: bench.py starts its own N ranks (one per GPU) when no launcher is around it
  python bench.py --gpus "$n" --steps "$steps" --warmup "$warm" --no-cpu-baseline > "$out/scale_N$n.json" 2> "$out/scale_N$n.err"
  python - "$out/scale_N$n.json" <<'PY' | tee -a "$out/scale_summary.txt"
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not line:
    print(sys.argv[1], "NO JSON LINE"); sys.exit(0)
d = json.loads(line[-1])
c = d.get("collective", {})
print(f"N={d['n_gpus']}: {d['value']:.1f} {d['unit']}  ({d['ms_per_step']:.3f} ms/step)  backend={c.get('backend', '-')} world={c.get('world_size', 1)} "
      f"devices={[r.get('pci_bus_id') or r.get('device') for r in c.get('ranks', [])]}")
PY
done
