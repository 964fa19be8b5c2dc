for n in 2 4 8; do CLIMA_BENCH_FORCE_DIST=1 CLIMA_BENCH_FAKE_SHARD=0,$n python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fake shard 0 of $n', '%.1f us/step' % (1e3*d['ms_per_step']), d['roofline'].get('kernel_us'))"; done
