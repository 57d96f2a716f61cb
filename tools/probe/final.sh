# end-of-round GPU pass (through gpurun):  bash tools/probe/final.sh <tag>
TAG=${1:-r06}
rm -f gpurun_out/parity_errors.jsonl
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -4
bash tools/profile_round.sh $TAG > gpurun_out/profile_round.log 2>&1
bash tools/pmc_round.sh > gpurun_out/pmc_round.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
