# same-box A/B of the three benchmark models: the library built from the tree against _lib/libpdt_kernels_prev.so
#   bash bench/ab_models.sh [models...]      (default: resnet50 vit_b_16 gpt2_medium)
set -o pipefail
mkdir -p gpurun_out/ab
models=${@:-resnet50 vit_b_16 gpt2_medium}
for m in $models; do
  for lib in new prev new2; do
    if [ $lib = prev ]; then export PDT_KERNELS_LIB=libpdt_kernels_prev.so; else unset PDT_KERNELS_LIB; fi
    timeout -k 10 240 python bench.py --model $m --steps 10 --warmup 3 --no-extras > gpurun_out/ab/${m}_$lib.log 2>&1 || { tail -5 gpurun_out/ab/${m}_$lib.log; exit 1; }
    python - "$m" "$lib" gpurun_out/ab/${m}_$lib.log <<'PY'
import json, sys
line = [l for l in open(sys.argv[3]) if l.startswith("{")][-1]
d = json.loads(line)
print(f"{sys.argv[1]:12s} {sys.argv[2]:5s} {d['value']:10.1f} {d['unit']:9s} {d['ms_per_step']:8.2f} ms", flush=True)
PY
  done
done
