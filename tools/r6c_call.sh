set -u
O=gpurun_out/r6c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 1
timeout -k 10 900 python tools/bf16_acceptance.py > $O/bf16_acceptance.json 2> $O/bf16_acceptance.err; echo "acceptance rc=$?"
tail -2 $O/bf16_acceptance.err
timeout -k 10 300 python tools/tower_ab.py ln_fold=1 ln_fold=0 --rounds 3 --reps 10 > $O/tower_ab.json 2> $O/tower_ab.err; cat $O/tower_ab.json
