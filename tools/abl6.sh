R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 tools/bench_legacy.py 2>&1 | tail -1 | cut -c1-260
for k in 1 3; do echo ABL $k; MCRAW_NOCHECK=1 MCRAW_LIB_PATH=$R/motioncam_decoder_amd/lib/ab/lib$k.so python3 tools/bench_legacy.py 2>&1 | tail -1 | cut -c1-260; done
