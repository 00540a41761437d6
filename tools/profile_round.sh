# Produces the committed profiles/ artefacts of a round on the GPU box: rocprofv3 kernel traces (default bench + serial mode),
# separate FETCH_SIZE / WRITE_SIZE PMC passes, and the default bench JSON.  Run: gpurun -- bash tools/profile_round.sh
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/prof8
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o run -- python $R/bench.py --no-cpu-baseline > $O/trace.log 2>&1
rocprofv3 --kernel-trace -d $O/serial -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --mode serial > $O/serial.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o run -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --mode serial > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o run -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --mode serial > $O/write.log 2>&1
cd $R
for d in trace serial fetch write; do f=$(find $O/$d -name "*.db" | head -1); echo "$d $f $(du -sh $f | cut -f1)"; done > $O/files.txt
python tools/rocpd_stats.py $(find $O/trace -name "*.db" | head -1) $O/kernel_stats_default.txt > /dev/null 2>&1
python tools/rocpd_stats.py $(find $O/serial -name "*.db" | head -1) $O/kernel_stats_serial.txt > /dev/null 2>&1
python tools/pmc_traffic.py $(find $O/fetch -name "*.db" | head -1) $(find $O/write -name "*.db" | head -1) $O/pmc_traffic.json 10 > $O/pmc.log 2>&1
find $O -name "*.db" -delete
python bench.py > $O/bench_default.log 2>&1
