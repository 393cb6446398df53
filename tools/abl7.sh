#!/bin/bash
# k7_records timing experiments (MCRAW_ABLATE_REC: 1 no unpack loop, 2 walk only, 3 no walk).
# Needs a library built with the experiment kernels:  MCRAW_DIAG=1 python -m motioncam_decoder_amd.build hip --force
for a in 0 1 2 3; do
  MCRAW_ABLATE_REC=$a python3 bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('abl_rec=$a', d['ms_per_step'], d['kernels_ms_per_step'], d['bit_exact'])"
done
