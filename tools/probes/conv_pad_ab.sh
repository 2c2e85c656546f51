set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for pad in 0 1 0 1; do
  echo "== pad=$pad"
  NANOFED_CONV_PATCH_PAD=$pad timeout -k 10 200 python -u tools/bench_conv.py --batch 512 --no-library --reps 100 > gpurun_out/conv_pad_ab_$pad.log 2>&1 || { tail -30 gpurun_out/conv_pad_ab_$pad.log; exit 1; }
  grep "^512" gpurun_out/conv_pad_ab_$pad.log
done
for pad in 0 1; do
  echo "== config 4 pad=$pad"
  NANOFED_CONV_PATCH_PAD=$pad timeout -k 10 400 python -u benchmarks/run_configs.py resnet18 --rounds 12 --warmup 2 > gpurun_out/conv_pad_cfg4_$pad.log 2>&1 || { tail -30 gpurun_out/conv_pad_cfg4_$pad.log; exit 1; }
  tail -1 gpurun_out/conv_pad_cfg4_$pad.log | cut -c1-160
done
