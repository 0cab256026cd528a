set -u
O=gpurun_out/r6b; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; echo "pytest rc=$rc"
[ $rc -eq 124 ] && exit 1
for i in 1 2; do for fo in 0 1; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-configs --steps 20 --warmup 5 --front-overlap $fo > $O/bench_fo${fo}_$i.json 2> $O/bench_fo${fo}_$i.err; rc=$?
  echo "bench fo=$fo run $i rc=$rc $(python -c "import json;d=json.load(open('$O/bench_fo${fo}_$i.json'));print(d['ms_per_step'], d['vit']['ms_per_batch'], d['value'])" 2>&1 | tail -1)"
  [ $rc -eq 124 ] && exit 1
done; done
timeout -k 10 900 python tools/bf16_acceptance.py > $O/bf16_acceptance.json 2> $O/bf16_acceptance.err; echo "acceptance rc=$?"
tail -2 $O/bf16_acceptance.err
