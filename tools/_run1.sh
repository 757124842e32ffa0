cd $GRAFT_REPO_ROOT
for k in 1 2; do
python tools/c5_probe.py 2>&1 | grep "window 64:" 
LPMP_ENGINE_SO=$PWD/build/exp/liblpmp_engine_r02.so python tools/c5_probe.py 2>&1 | grep "window 64:"
done
