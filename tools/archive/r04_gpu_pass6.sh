# DEV TOOL (GPU box), round 4 pass 6: what the seed-chain validation finds per workload (tests/tools/validate_stats.py), how a launch of the persistent
# kernel ends with tail help shipped (-DLHIP_WAVE_TIMES build, tests/tools/wave_tail.py), phase cycles of the scratch-free g_quant.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pass6; mkdir -p $O
cd $R
timeout 200 python tests/tools/validate_stats.py 2>&1 | grep -v "amdgpu.ids" | tee $O/validate_stats.txt
timeout 240 python tests/tools/wave_tail.py 16384 100000 2>&1 | grep -v "amdgpu.ids" | tee $O/wave_tail.txt
timeout 90 python tests/tools/phase_prof.py 20000 > $O/quant_phase_cycles.txt 2>&1; grep -v "^   [cnq]_" $O/quant_phase_cycles.txt | head -40
