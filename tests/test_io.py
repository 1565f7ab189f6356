"""snapshots keep the reference's AoS array bit for bit; VTK export of cell data."""
import numpy as np


def test_snapshot_roundtrip_fv_and_dg(tmp_path):
    from exahype_amd.io import load_snapshot, save_snapshot
    Q = np.sin(3.141 * np.arange(360) / 360).reshape(1, 6, 6, 10)            # the reference's test array
    save_snapshot(tmp_path / "q.npz", Q, "fv", time=0.25, dim=2, patch_size=4, halo_size=1, n_real=5, n_aux=5)
    data, h = load_snapshot(tmp_path / "q.npz")
    assert np.array_equal(data, Q) and h["kind"] == "fv" and h["time"] == 0.25 and h["patch_size"] == 4
    u = np.random.default_rng(0).random((2, 3, 4, 4, 5))
    save_snapshot(tmp_path / "u.npz", u, "dg", dim=2, N=4, n_vars=5, dx=[0.5, 1 / 3])
    data, h = load_snapshot(tmp_path / "u.npz")
    assert np.array_equal(data, u) and h["N"] == 4 and h["shape"] == [2, 3, 4, 4, 5]


def test_cell_averages_and_vtk(tmp_path):
    from exahype_amd.io import dg_cell_averages, write_vtk
    from oracle.dg_operators import operators
    ops = operators(3)
    u = np.ones((2, 3, 3, 3, 2)) * np.array([1.5, -2.0])
    u[1, 2] *= 2
    avg = dg_cell_averages(u, ops["w"])
    assert avg.shape == (2, 3, 2) and np.allclose(avg[0, 0], [1.5, -2.0]) and np.allclose(avg[1, 2], [3.0, -4.0])
    write_vtk(tmp_path / "f.vtk", avg, [0.5, 1 / 3], names=["rho", "m"])
    txt = open(tmp_path / "f.vtk").read()
    assert "DIMENSIONS 3 4 2" in txt and "CELL_DATA 6" in txt and "SCALARS rho double 1" in txt
    vals = txt.split("SCALARS rho double 1\nLOOKUP_TABLE default\n")[1].split("SCALARS")[0].split()
    assert len(vals) == 6 and abs(float(vals[0]) - 1.5) < 1e-14 and abs(float(vals[5]) - 3.0) < 1e-14
