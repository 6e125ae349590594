cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
( time DFH_PARITY_RECORD=$O/parity.json DFH_PARITY_RECORD_STEPS=$O/parity_steps.json timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300; grep -E "^E |^FAILED" $O/pytest_gpu.log | head -30
bash tools/gpu_r05h.sh
