#!/opt/conda/bin/python3.9
"""Golden .flo5 files written by the REAL h5py / libhdf5 exactly the way the reference writes Spring flow
(/root/reference/core/utils/frame_utils.py:31-47: one dataset 'flow', gzip level 5, h5py's automatic chunking), so that
streamflow_amd/flo5.py's reader is checked against bytes it did not produce.

Run in the build container with the conda interpreter that has h5py (this repo's own python has none):
    /opt/conda/bin/python3.9 tests/golden/make_flo5_golden.py
Outputs (committed, data only): tests/golden/flo5/*.flo5 and tests/golden/flo5/expected.npz (the arrays that were written,
with NaNs marking invalid pixels like Spring does).  The second part re-reads files written by OUR writer with h5py
(writer parity); tests/test_flow_io_cpu.py repeats that check when this interpreter is present."""
import os
import sys

import h5py
import numpy as np

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "flo5")
os.makedirs(here, exist_ok=True)
rng = np.random.default_rng(20260104)
cases = {"tiny_1x1": (1, 1), "ragged_37x53": (37, 53), "upright_70x9": (70, 9), "wide_136x240": (136, 240)}
exp = {}
for name, (h, w) in cases.items():
    flow = (rng.standard_normal((h, w, 2)) * 20).astype(np.float32)
    if h * w > 4:
        flow[rng.random((h, w)) < 0.1] = np.nan                 # Spring marks invalid pixels with NaN
    with h5py.File(os.path.join(here, name + ".flo5"), "w") as f:  # frame_utils.py:46-47, verbatim call
        f.create_dataset("flow", data=flow, compression="gzip", compression_opts=5)
    exp[name] = flow
# a float64 dataset, an uncompressed contiguous one and a chunked + shuffled one: layouts the reader also accepts
f64 = rng.standard_normal((5, 7, 2))
with h5py.File(os.path.join(here, "f64_contiguous.flo5"), "w") as f:
    f.create_dataset("flow", data=f64)
exp["f64_contiguous"] = f64
sh = (rng.standard_normal((33, 40, 2)) * 3).astype(np.float32)
with h5py.File(os.path.join(here, "shuffle_chunks.flo5"), "w") as f:
    f.create_dataset("flow", data=sh, compression="gzip", compression_opts=9, shuffle=True, chunks=(8, 16, 2))
exp["shuffle_chunks"] = sh
np.savez_compressed(os.path.join(here, "expected.npz"), **exp)
print("wrote", sorted(os.listdir(here)), "h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)

# writer parity: files written by streamflow_amd.flo5 (paths given on the command line) must open in h5py
for path in sys.argv[1:]:
    with h5py.File(path, "r") as f:
        d = f["flow"]
        a = d[()]
        print("READBACK", path, a.shape, a.dtype, d.compression, d.compression_opts, d.chunks,
              float(np.nansum(a.astype(np.float64))))
