cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/final
PART=a bash scripts/final_measure.sh 2>&1 | grep -v "^+" | tail -26 | cut -c1-2000
