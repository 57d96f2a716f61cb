rm -f gpurun_out/parity_errors.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/profile_round.sh r03 > gpurun_out/profile_round.log 2>&1
bash tools/pmc_round3.sh > gpurun_out/pmc_round3.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
