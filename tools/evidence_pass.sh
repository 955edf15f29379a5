cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_v34
python -m pytest tests -m gpu -q 2>&1 | tail -130 > gpurun_out/r06_gputest_final.log; tail -3 gpurun_out/r06_gputest_final.log
tools/profile_round.sh v34 > gpurun_out/r06_profile_round.log 2>&1
tools/profile_kernels.sh v34 > gpurun_out/r06_profile_kernels.log 2>&1
python tools/pmc_summary.py v34 r06 > gpurun_out/r06_pmc_summary.log 2>&1
tools/rollout_grid.sh > $P/rollout_grid.log 2>&1
python tools/throughput_time.py > $P/throughput.json 2>/dev/null
python tools/determinism_probe.py > $P/determinism_probe.txt 2>&1
python tools/determinism_probe.py --robot icub > $P/determinism_probe_three_per_cu.txt 2>&1
python tools/phase_profile.py --batch 256 --noise 0.5 > $P/phase_b256.txt 2>&1
python tools/loop_waves.py > $P/loop_waves.txt 2>&1
python tools/iters_floor.py > $P/iters_floor.txt 2>&1
python tools/layout_crosscheck.py > $P/layout_crosscheck.txt 2>&1
tools/occ3_pmc.sh v34 > gpurun_out/r06_occ3.log 2>&1
python tools/occ3_pmc_summary.py v34 r06 > gpurun_out/r06_occ3_summary.log 2>&1
python tools/active_set_diag.py > $P/active_set_diag.txt 2>&1
python tools/residency_sweep.py > $P/residency_sweep.txt 2>&1
ls $P | wc -l
tail -1 $P/bench.json | cut -c1-400
