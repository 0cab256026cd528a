set -u
O=gpurun_out/r6e; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_vit_gpu.py -q -m gpu -x -k "attention or ab_hooks or cls_only" > $O/pytest_vit.log 2>&1; rc=$?; tail -2 $O/pytest_vit.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 400 python tools/tower_ab.py attn_nt=0 attn_nt=1 attn_nt=1,qkv_layout=1 attn_nt=0,qkv_layout=1 --rounds 4 --reps 10 > $O/tower_ab_nt.json 2> $O/tower_ab_nt.err; cat $O/tower_ab_nt.json
