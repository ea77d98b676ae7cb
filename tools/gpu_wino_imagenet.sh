timeout 900 python -m pytest tests/test_conv_wino_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -6
timeout 300 python tools/conv_bench.py --batch 64 --imagenet --algo winograd 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['N'], r['Ci'], r['HW'], r['algo'], 'fwd', r.get('fwd_us'), 'dgrad', r.get('dgrad_us'), 'lib', r.get('fwd_us_lib'), r.get('dgrad_us_lib'), 'err %.1e %.1e' % (r.get('fwd_err', 0), r.get('dgrad_err', 0)))"
