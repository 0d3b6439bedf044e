cd /tmp && export TMPDIR=/tmp
for spp in 128 1024; do
rm -rf /tmp/kt_$spp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$spp -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --spp $spp > $GRAFT_REPO_ROOT/gpurun_out/kt_$spp.log 2>&1
cp $(find /tmp/kt_$spp -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/kt_${spp}_kernel_stats.csv
cp $(find /tmp/kt_$spp -name "*kernel_trace.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/kt_${spp}_kernel_trace.csv
done
