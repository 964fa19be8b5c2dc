#!/usr/bin/env python3
"""Generate tests/golden/datadir_c/: a tiny opacity-data directory in the schema of the reference's
loaders (src/radtran/clima_radtran_types_create.f90:734-1468) whose HDF5 files are written by
tests/golden/h5pack.c -- the HDF5 C library's own dataset-creation path with CHUNKED layout, SHUFFLE +
DEFLATE filters and, for one k-table and the Mie tables, FLOAT32 storage -- not by clima_amd/h5lite.py.
Reading it with clima_amd/data_loader.py is therefore not a round trip through one writer/reader pair.

Also writes tests/golden/datadir_c_written.npz: every array that went into the files (float64, before
any float32 rounding), from which the tests build their expected tables WITHOUT the loader.

Run in the build container (needs gcc + the HDF5 C library under /opt/conda):
    python tests/golden/make_datadir_c.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
from clima_amd import synthetic as S  # noqa: E402
from datadir_fixture import write_datadir  # noqa: E402

F32 = {("kdistributions/CO2.h5", "log10k"), ("aerosol_xsections/khare1984/mie_khare1984.h5", "w0"),
       ("aerosol_xsections/khare1984/mie_khare1984.h5", "qext"), ("aerosol_xsections/khare1984/mie_khare1984.h5", "g0")}


def main():
    out = os.path.join(HERE, "datadir_c")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    tool = os.path.join(tempfile.gettempdir(), "h5pack")
    subprocess.check_call(["gcc", "-O2", "-I/opt/conda/include", os.path.join(HERE, "h5pack.c"), "-o", tool,
                           "-L/opt/conda/lib", "-lhdf5", "-Wl,-rpath,/opt/conda/lib"])
    stored = {}

    def h5write(path, datasets):
        rel = os.path.relpath(path, out)
        with tempfile.TemporaryDirectory() as td:
            lines = []
            for i, (name, arr) in enumerate(datasets.items()):
                a = np.ascontiguousarray(arr, dtype=np.float64)
                raw = os.path.join(td, "d%d.raw" % i)
                a.tofile(raw)
                dims = a.shape if a.ndim else (1,)
                # chunks smaller than the dataset in every dimension that allows it: several chunks per dataset
                chunk = [max(1, (d + 1) // 2) for d in dims]
                typ = "f32" if (rel, name) in F32 else "f64"
                lines.append("%s %s %d %s %s %d %s" % (name, typ, len(dims), " ".join(map(str, dims)),
                                                         " ".join(map(str, chunk)), 4, raw))
                stored[rel + ":" + name] = a
            spec = os.path.join(td, "spec.txt")
            with open(spec, "w") as f:
                f.write("\n".join(lines) + "\n")
            subprocess.check_call([tool, path, spec])

    tb = S.make_tables(nw=16, ng=8, nP=5, nT=4, seed=21)
    written = write_datadir(out, tb, rng=np.random.default_rng(5), h5write=h5write)
    flat = {"wavl": tb.wavl, "species": np.array(tb.species_names), "particles": np.array(tb.particle_names)}
    for k, v in stored.items():
        flat["file:" + k] = v
    ws, flux = written["star"]
    flat["star_w"], flat["star_f"] = ws, flux
    np.savez_compressed(os.path.join(HERE, "datadir_c_written.npz"), **flat)
    size = sum(os.path.getsize(os.path.join(b, f)) for b, _, fs in os.walk(out) for f in fs)
    print("wrote", out, "%d bytes in %d files" % (size, sum(len(fs) for _, _, fs in os.walk(out))))


if __name__ == "__main__":
    main()
