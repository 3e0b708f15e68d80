cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest_v3.log 2>&1 || { tail -40 gpurun_out/r03_gputest_v3.log; exit 1; }
tail -2 gpurun_out/r03_gputest_v3.log
python __graft_entry__.py smoke 2>&1 | tail -1
# the reference's CLI, end to end: synthetic reader, then an array file through the npz reader with the blob assembled on the device
python bin/uresnet.py train -mn uresnet_sparse -io synthetic_sparse -ss 128 -uf 16 -uns 3 -nc 5 -bs 2 -mbs 2 -np 3000 -it 6 -rs 2 --gpus 0 -dkeys data,label -sd 1 2>&1 | grep "Iter\|loss" | tail -4
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from uresnet_pytorch_amd.iotools import array_io
from uresnet_pytorch_amd.iotools.synthetic import generate_event
ev = []
for s in range(6):
    c, v, l = generate_event(s, 128, 2500 + 100 * s)
    ev.append({'voxels': c, 'feature': v, 'label': l})
array_io.write_sparse_npz('/tmp/events.npz', ev)
PY
python bin/uresnet.py train -mn uresnet_sparse -io npz_sparse -if /tmp/events.npz -iod -cw -ss 128 -uf 16 -uns 3 -nc 5 -bs 2 -mbs 2 -it 6 -rs 2 --gpus 0 -dkeys data,label -sd 1 -wp /tmp/ck/snap -chks 3 -cmp 2>&1 | grep "Iter\|loss" | tail -4
ls /tmp/ck/
python bin/uresnet.py inference -mn uresnet_sparse -io npz_sparse -if /tmp/events.npz -ss 128 -uf 16 -uns 3 -nc 5 -bs 1 -mbs 1 -it 3 -rs 1 --gpus 0 -dkeys data,label -sd 1 -mp /tmp/ck/snap-5.ckpt -of /tmp/pred.npz -ld /tmp/logs 2>&1 | grep "Iter\|loss\|Restoring\|Done" | tail -5
python -c "import numpy as np; z=np.load('/tmp/pred.npz'); print(sorted(z.files)[:4], z['prediction/0'].shape)"
