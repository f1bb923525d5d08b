# rocprofv3 kernel trace + stats of the config-5 stand-in epoch and its kernel-family account
#   bash scripts/rocprof_tgn_epoch.sh [tag]   (on the GPU box via gpurun; the program directly after --)
export TMPDIR=/tmp
TAG=${1:-r06}
mkdir -p /tmp/prof gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o tgn -- python3 examples/tgn_epoch.py --epochs 2 --max-batches 400 > /tmp/prof/tgn_epoch.log 2>&1
grep "^{" /tmp/prof/tgn_epoch.log | tail -1 > gpurun_out/$TAG/tgn_epoch_profiled.json
cp /tmp/prof/tgn_kernel_stats.csv gpurun_out/$TAG/tgn_epoch_kernel_stats.csv
python3 scripts/tgn_epoch_kernel_account.py gpurun_out/$TAG/tgn_epoch_kernel_stats.csv gpurun_out/$TAG/tgn_epoch_profiled.json > gpurun_out/$TAG/tgn_epoch_kernel_summary.json
head -60 gpurun_out/$TAG/tgn_epoch_kernel_summary.json
# the unprofiled epoch for the record
python3 examples/tgn_epoch.py --epochs 3 > gpurun_out/$TAG/tgn_epoch_mag_shaped.json 2>/dev/null
cut -c1-600 gpurun_out/$TAG/tgn_epoch_mag_shaped.json
