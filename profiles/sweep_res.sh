#!/bin/bash
# frame time vs ray count on one GPU (what one rank of an N-GPU ray-sharded frame sees): 800^2/N rays for N = 1, 2, 4, 8
for r in 800 566 400 283; do
  python bench.py --res $r --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('res', $r, 'ms/frame %.3f' % d['ms_per_step'], 'Msamples/s %.1f' % (d['value']/1e6), 'samples/frame', d['config']['rendered_samples_per_step'])"
done
