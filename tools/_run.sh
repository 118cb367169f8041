export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/pq; timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d /tmp/pq -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1
python tools/pmc_summary.py $(find /tmp/pq -name "*counter_collection.csv") > gpurun_out/pmc_sq.csv
wc -l gpurun_out/pmc_sq.csv; tail -3 gpurun_out/pmc_sq.log
