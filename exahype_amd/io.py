"""Snapshot I/O in the reference's array layout.

The only "wire format" the reference implies is the AoS patch array of `exahype/printers/CPPPrinter.py:247-261`
(row-major, variable fastest, halo included for FV patches).  `save_snapshot` / `load_snapshot` keep exactly that
array (numpy .npz, no pickling) together with the geometry needed to interpret it; `write_vtk` writes a legacy-VTK
structured-points file for visual checks (the reference only #includes Peano's `observers/PlotSolution.h`,
`CPPPrinter.py:328`)."""
import json

import numpy as np

FORMAT = 1


def save_snapshot(path, array, kind, time=0.0, **meta):
    """kind: "fv" (Q[patch.., i, j, (k,) var], halo included) or "dg" (u[cell.., node.., var]).
    meta: dim, patch_size/halo_size/n_real/n_aux (fv) or N/n_vars/dx (dg), grid."""
    a = np.ascontiguousarray(np.asarray(array, dtype=np.float64))
    header = dict(format=FORMAT, kind=kind, time=float(time), shape=list(a.shape), **meta)
    np.savez(path, data=a, header=np.frombuffer(json.dumps(header).encode(), dtype=np.uint8))


def load_snapshot(path):
    with np.load(path, allow_pickle=False) as z:
        header = json.loads(bytes(z["header"]).decode())
        data = z["data"]
    if header.get("format") != FORMAT or list(data.shape) != header["shape"]:
        raise ValueError("not an exahype_amd snapshot: %s" % path)
    return data, header


def dg_cell_averages(u, w):
    """Cell averages of a DG field u[grid.., nodes.., var] with the Gauss-Legendre weights w."""
    dim = (u.ndim - 1) // 2
    out = np.asarray(u)
    for _ in range(dim):
        out = np.tensordot(w, out, axes=([0], [dim]))          # contracts the first remaining node axis
    return out


def write_vtk(path, field, spacing, names=None):
    """Legacy VTK STRUCTURED_POINTS, cell data.  field[g0, g1, (g2,) var] (e.g. FV interiors or DG cell averages)."""
    f = np.asarray(field, dtype=np.float64)
    dim = f.ndim - 1
    g = list(f.shape[:dim]) + [1] * (3 - dim)
    sp = list(spacing) + [1.0] * (3 - len(spacing))
    names = names or ["var%d" % v for v in range(f.shape[-1])]
    with open(path, "w") as out:
        out.write("# vtk DataFile Version 3.0\nexahype_amd snapshot\nASCII\nDATASET STRUCTURED_POINTS\n")
        out.write("DIMENSIONS %d %d %d\nORIGIN 0 0 0\nSPACING %r %r %r\n" % (g[0] + 1, g[1] + 1, g[2] + 1, sp[0], sp[1], sp[2]))
        out.write("CELL_DATA %d\n" % (g[0] * g[1] * g[2]))
        for v, name in enumerate(names):
            out.write("SCALARS %s double 1\nLOOKUP_TABLE default\n" % name)
            vals = f[..., v].reshape(g)
            # VTK wants x fastest: our axis 0 is x -> transpose to (z, y, x) order
            out.write("\n".join(repr(float(x)) for x in np.transpose(vals, (2, 1, 0)).ravel()) + "\n")
