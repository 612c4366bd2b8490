mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.readline()); print(d["config"]["half_bandwidth"], round(d["ms_per_step"],4), round(d["roofline"]["launch_ms"],4), round(d["roofline"]["frac"],4))'
run() { python bench.py --steps 40 --warmup 5 --bandwidth $1 --no-cpu-baseline --no-sweep 2>&1 | grep metric | python -c "$P"; }
{
for rep in 1 2; do
for spw in 3 4 5 6 7; do echo -n "w4096 spw=$spw "; SL_BAND_SPW=$spw run 4096; done
for spw in 1 2 3 4; do echo -n "w512 spw=$spw "; SL_BAND_SPW=$spw run 512; done
echo -n "w4096 default "; run 4096
echo -n "w4096 ntstore "; SUBLINEAR_HIP_LIB=$PWD/sublinear_time_solver_amd/libsublinear_hip_nt.so run 4096
echo -n "w512 default "; run 512
echo -n "w512 ntstore "; SUBLINEAR_HIP_LIB=$PWD/sublinear_time_solver_amd/libsublinear_hip_nt.so run 512
done
} > gpurun_out/sweep5.txt 2>&1
cat gpurun_out/sweep5.txt
