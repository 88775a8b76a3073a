cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "amdgpu.ids" | tail -8
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
PART=a bash scripts/final_measure.sh 2>&1 | grep -v "^+" | tail -22
