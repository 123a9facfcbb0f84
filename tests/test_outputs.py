"""TUM trajectory lines and `record` tables byte-for-byte against the oracle, which renders them with
the reference's own iostream calls (oracle/io_formats.h)."""
import numpy as np


def test_tum_lines_byte_exact(liw, synth, pyoracle, tmp_path):
    prm = synth.office_params()
    rng = np.random.default_rng(0)
    path = tmp_path / "fornt_end.txt"          # the reference's file name (trajectory.cpp:61)
    expect = "#Time px py pz qx qy qz qw\n"
    with liw.outputs.TumWriter(path, prm) as w:
        t = 1560000000.123456
        for k in range(40):
            p = rng.normal(0, 20, 3)
            q = rng.normal(0, 1.0, 3) * (3.0 if k % 5 == 0 else 1.0)     # incl. rotations with trace <= 0 (all quaternion branches)
            t += 0.1
            assert w.append(t, p, q) == 0
            expect += pyoracle.tum_line(prm["T_imu_to_wheel"], True, t, p, q)
        assert w.append(t, p, q) == -1        # time did not increase: flagged, still written (reference logs an error)
        expect += pyoracle.tum_line(prm["T_imu_to_wheel"], True, t, p, q)
    got = open(path).read()
    assert got == expect
    cols = got.splitlines()[1].split(" ")
    assert len(cols) == 8 and all(len(c.split(".")[1]) == 10 for c in cols)


def test_tum_pose_is_base_pose(liw, synth):
    prm = synth.office_params()
    p, q = np.array([1.0, 2.0, 0.1]), np.array([0.02, -0.01, 1.3])
    v = liw.outputs.tum_pose(prm, p, q)
    T_iw = np.array(synth.normalize_extrinsic(prm["T_imu_to_wheel"])).reshape(4, 4)
    T = np.eye(4)
    T[:3, :3] = synth.exp_so3(q)
    T[:3, 3] = p
    Tb = T @ T_iw
    assert np.abs(v[:3] - Tb[:3, 3]).max() < 1e-5
    x, y, z, w = v[3:]
    assert abs(x * x + y * y + z * z + w * w - 1.0) < 1e-5
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    assert np.abs(R - Tb[:3, :3]).max() < 1e-5


def test_record_tables_byte_exact(liw, pyoracle, tmp_path):
    rng = np.random.default_rng(1)
    r, o = liw.outputs.Record(), pyoracle.OracleRecord()
    for name, cnt, scale in (("solve", 57, 12000), ("marginalization", 57, 900), ("one", 1, 5)):
        for v in rng.integers(1, scale, cnt):
            r.add_time(name, v); o.add_time(name, v)
    for name in ("laser_match_size", "iterations"):
        for v in rng.integers(0, 3000, 23):
            r.add_record(name, v); o.add_record(name, v)
    assert r.format() == o.dump()
    assert "| solve | 57 | " in r.format() and r.format().startswith("time_recorder\nsize of total record type:3\n")
    assert r.write(tmp_path / "traj.md") == 0 and open(tmp_path / "traj.md").read() == o.dump()
    # empty recorder and live timing
    e = liw.outputs.Record()
    assert e.format() == pyoracle.OracleRecord().dump()
    e.begin_record(); e.begin_record()
    assert e.end_record("inner") >= 0 and e.end_record("outer") >= 0 and e.end_record("unbalanced") == 0
    assert "| inner | 1 | " in e.format() and "| outer | 1 | " in e.format() and "unbalanced" not in e.format()
