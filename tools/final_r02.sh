# end-of-round run on one box: the full GPU suite, smoke, then the judged profiles (tools/collect_profiles.sh)
set -x
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r02_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r02_final_tests.log 2>&1
bash tools/collect_profiles.sh r02
cat gpurun_out/r02_final_tests.log
