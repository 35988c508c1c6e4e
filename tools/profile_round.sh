set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
rm -rf gpurun_out/prof_round gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_round -o p -- python3 bench.py --steps 2 --warmup 1 --no-extras > gpurun_out/bench_prof_b.json 2> gpurun_out/bench_prof_b.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 8 --seconds 2049 > gpurun_out/pmcf.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 8 --seconds 2049 > gpurun_out/pmcw.out 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o p -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 8 --seconds 2049 > gpurun_out/pmcm.out 2>&1
ls gpurun_out/prof_round gpurun_out/pmc_fetch gpurun_out/pmc_mfma
