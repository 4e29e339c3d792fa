cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4/pytest.log
grep -n "passed\|failed\|rc=" gpurun_out/r4/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
