#!/bin/bash
# tools/probe/variant_ab.sh <out dir> <rounds> <variant names...>: the library against probe builds of it (tools/probe/build_variant.sh),
# alternating processes on ONE box; prints ms per 256-image forward (3 rounds x 10 forwards each process)
O=$1; R=$2; shift 2; mkdir -p $O
for i in $(seq 1 $R); do
  python tools/tower_ab.py ln_fold=1 --rounds 3 --reps 10 > $O/base_$i.json 2>/dev/null; python -c "import json;d=json.load(open('$O/base_$i.json'));print('library'.ljust(12), d['variants']['ln_fold=1']['ms'])"
  for v in "$@"; do
    python tools/tower_ab.py ln_fold=1 --rounds 3 --reps 10 --lib tools/probe/variant/$v/libmi355clip.so > $O/${v}_$i.json 2>/dev/null; python -c "import json;d=json.load(open('$O/${v}_$i.json'));print('$v'.ljust(12), d['variants']['ln_fold=1']['ms'])"
  done
done
