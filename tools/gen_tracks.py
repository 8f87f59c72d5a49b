"""Regenerate mpopis_amd/data/*_sf20.csv: the reference's centre-line data files sub-sampled exactly as
Track(infile; sample_factor=20) does (x[1:20:end], src/envs/car_racing_tracks/car_racing_tracks.jl:21-23).
Development-time only (reads /root/reference, which does not exist on the GPU box); the outputs are data
fixtures committed to the repo.  Values are written with repr() so they round-trip bit-exactly."""
import glob
import os
import numpy as np

SRC = "/root/reference/src/envs/car_racing_tracks"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpopis_amd", "data")

for f in sorted(glob.glob(os.path.join(SRC, "*.csv"))):
    d = np.loadtxt(f, delimiter=",")
    assert d.ndim == 2 and d.shape[1] == 2
    sub = d[::20]
    out = os.path.join(DST, os.path.basename(f)[:-4] + "_sf20.csv")
    with open(out, "w") as fh:
        for x, y in sub:
            fh.write("%r,%r\n" % (float(x), float(y)))
    back = np.loadtxt(out, delimiter=",")
    assert np.array_equal(back, sub)
    print(os.path.basename(out), len(sub), "points")
