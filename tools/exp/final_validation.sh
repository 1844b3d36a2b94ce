cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06_final_gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r06_final_gputest.txt 2>&1
tail -8 gpurun_out/r06_final_gputest.txt
