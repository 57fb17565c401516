mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/profile_round2.sh r04 > gpurun_out/profile_r04.log 2>&1; echo "profile rc=$?"; tail -25 gpurun_out/profile_r04.log
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench rc=$?"
