# end-of-round run on one box: the full GPU suite, smoke, the judged profiles (tools/collect_profiles.sh), the step timeline
set -x
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r02_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -E "smoke|Error|error" >> gpurun_out/r02_final_tests.log
bash tools/collect_profiles.sh r02
bash tools/r02_timeline.sh > /dev/null 2>&1
cat gpurun_out/r02_final_tests.log
