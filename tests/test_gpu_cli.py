"""GPU: the spotfinder driver end to end -- stdout phrases, --pipe_fd JSON lines, files and exit
codes of the reference's CLI contract (spotfinder/spotfinder.cc, tests/test_spotfinder.py:26-29,
tests/3d_connected_components.sh:48-62), with expected values from the oracle."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "fast-feedback-service_amd", "bin")
SPOTFINDER = os.path.join(BIN, "spotfinder")
TOOL = os.path.join(BIN, "ffs_hosttool")

pixels_match_regex = r"image\s+(\d+).*?(\d+)\s+strong pixels"            # reference tests/test_spotfinder.py:26
spots_match_regex = r"Calculated\s+(\d+)\s+spots"
min_spot_size_regex = r"Filtered\s+(\d+)\s+spots with size < (\d+) pixels"
max_separation_regex = r"Filtered\s+(\d+)\s+spots with peak-centroid distance > 2"


def strip_ansi(t):
    return re.sub(r"\x1b\[[0-9;]*m", "", t)


def tiny_frames(n, sweep=False, seed=7):
    from ffs_amd import synth
    p = synth.params(300, 200, np.uint16, seed=seed, background=2.0, n_spots=40, sigma=(0.8, 1.6),
                     peak=(30.0, 5000.0), max_value=65535,
                     n_frames=n if sweep else 0, sigma_z=(0.5, 2.0) if sweep else (0.0, 0.0))
    return synth.frames(p, range(n), threads=2)


def run_with_pipe(argv, cwd):
    r, w = os.pipe()
    proc = subprocess.Popen([SPOTFINDER, *argv, "--pipe_fd", str(w)], pass_fds=[w], cwd=cwd,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    os.close(w)
    out, err = proc.communicate(timeout=300)
    with os.fdopen(r) as f:
        lines = [l for l in f.read().split("\n") if l]
    return proc.returncode, out, err, lines


def expected(frames, mask, min_spot_size=3):
    from oracle import oracle as O
    exp = []
    for img in frames:
        strong = O.dispersion(img, mask)
        cc = O.cc2d(strong, img, min_spot_size)
        refl = O.cc2d_reflections(cc.k, cc.intensity, img.shape[1], img.shape[0], min_spot_size, 2.0)
        exp.append((cc, refl))
    return exp


@pytest.mark.parametrize("threads,batch", [(1, 1), (3, 2), (2, 4)])
def test_stills_json_lines_and_stdout(tmp_path, threads, batch):
    N = 7
    rc, out, err, lines = run_with_pipe(["synth:tiny:%d" % N, "--threads", str(threads), "--batch", str(batch),
                                         "--output-for-index"], tmp_path)
    assert rc == 0 and not err, err
    frames = tiny_frames(N)
    exp = expected(frames, np.ones((200, 300), np.uint8))
    got = {}
    for l in lines:
        # nlohmann dump(): alphabetical keys, no spaces
        assert list(json.loads(l).keys()) == sorted(json.loads(l).keys()) and ", " not in l
        j = json.loads(l)
        got[j["file-number"]] = j
    assert sorted(got) == list(range(N))
    # threads == 1 prints the reference's multi-line timing block (spotfinder.cc:1056-1076); the
    # reference's own regex targets the one-line form printed with several threads (:1078-1085)
    flags = re.S if threads == 1 else 0
    found = dict((int(a), int(b)) for a, b in re.findall(pixels_match_regex, strip_ansi(out), flags))
    for i, (cc, refl) in enumerate(exp):
        j = got[i]
        assert j["file"] == "synth:tiny:%d" % N
        assert j["num_strong_pixels"] == cc.num_strong_pixels == found[i]
        assert j["n_spots_total"] == len(cc.boxes)
        want = np.stack([refl.reflections["com_x"], refl.reflections["com_y"], refl.reflections["com_z"]], 1).reshape(-1)
        np.testing.assert_allclose(np.array(j["spot_centers"], np.float32), want, rtol=0, atol=1e-6)
    assert f"{N} images in" in out


def test_rotation_3d_output(tmp_path):
    N = 8
    rc, out, err, lines = run_with_pipe(["synth:tinysweep:%d" % N, "--threads", "2", "--batch", "3", "--writeout",
                                         "--min-spot-size-3d", "4"], tmp_path)
    assert rc == 0 and not err, err
    from oracle import oracle as O
    frames = tiny_frames(N, sweep=True)
    mask = np.ones((200, 300), np.uint8)
    exp = expected(frames, mask)
    want = O.cc3d([(cc.k, cc.intensity) for cc, _ in exp], 300, 200, 4, 2.0)
    txt = strip_ansi(out)
    assert "Dataset type: Rotation set" in txt
    assert int(re.search(spots_match_regex, txt).group(1)) == want.n_calculated
    m = re.search(min_spot_size_regex, txt)
    assert (int(m.group(1)), int(m.group(2))) == (want.n_filtered_size, 4)
    if want.n_filtered_sep:
        assert int(re.search(max_separation_regex, txt).group(1)) == want.n_filtered_sep
    assert f"Found {len(want.reflections)} spots" in txt
    got = open(tmp_path / "3d_reflections.txt").read().strip().split("\n")
    assert len(got) == len(want.reflections) > 3
    for line, r in zip(got, want.reflections):
        m = re.match(r"X: \[(\d+), (\d+)\] Y: \[(\d+), (\d+)\] Z: \[(\d+), (\d+)\] COM: \(([^,]+), ([^,]+), ([^)]+)\)", line)
        assert m, line
        assert [int(m.group(i)) for i in range(1, 7)] == [r["x_min"], r["x_max"], r["y_min"], r["y_max"], r["z_min"], r["z_max"]]
        # iostream default formatting = 6 significant digits (spotfinder.cc:1138-1147)
        for g, w in zip(m.groups()[6:], (r["com_x"], r["com_y"], r["com_z"])):
            assert g == "%g" % w
    # --writeout also lists strong pixels per image ("{:4d}, {:4d}", spotfinder.cc:985-993)
    px = open(tmp_path / "pixels_00000.txt").read().rstrip("\n").split("\n")
    assert len(px) == exp[0][0].num_strong_pixels
    ys, xs = np.divmod(exp[0][0].k.astype(np.int64), 300)
    assert px[0] == "%4d, %4d" % (xs[0], ys[0])
    assert (tmp_path / "image_00000.png").read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"


def test_shm_and_cbf_readers(tmp_path):
    """Frames written as an Eiger-stream directory (bitshuffle-LZ4) and as miniCBF (byte-offset)
    give the same answers as the raw synthetic source."""
    N = 4
    spec = "synth:tiny:%d" % N
    shm = tmp_path / "shm"
    assert subprocess.run([TOOL, "mkshm", spec, str(shm)]).returncode == 0
    assert subprocess.run([TOOL, "mkcbf", spec, str(tmp_path / "img_")]).returncode == 0
    exp = expected(tiny_frames(N), np.ones((200, 300), np.uint8))
    det = json.dumps({"pixel_size_x": 0.075, "pixel_size_y": 0.075, "beam_center_x": 11.25, "beam_center_y": 7.5,
                      "distance": 300.0})
    for argv in (["%s" % shm, "--threads", "2"],
                 ["%s" % shm, "--threads", "2", "--batch", "3", "--cpu-decode"],
                 [str(tmp_path / "img_####.cbf"), "--images", str(N), "--start-index", "1", "--wavelength", "0.976",
                  "--detector", det]):
        rc, out, err, lines = run_with_pipe(argv, tmp_path)
        assert rc == 0 and not err, (out, err)
        got = {json.loads(l)["file-number"]: json.loads(l) for l in lines}
        for i, (cc, _) in enumerate(exp):
            assert got[i]["num_strong_pixels"] == cc.num_strong_pixels
            assert got[i]["n_spots_total"] == len(cc.boxes)
            assert got[i]["file"] == argv[0]


@pytest.mark.parametrize("layout", ["vds-links", "plain"])
def test_hdf5_nxmx_reader(tmp_path, layout):
    """An NXmx master (virtual dataset over externally linked data files, or one chunked dataset)
    gives the oracle's answers; wavelength/geometry come from the file so --dmin works unaided."""
    N = 5
    master = tmp_path / "coll_master.h5"
    p = subprocess.run([TOOL, "mkh5", "synth:tiny:%d" % N, str(master), layout, "2"], capture_output=True, text=True)
    if "HDF5-enabled" in p.stdout:
        pytest.skip("built without HDF5")
    assert p.returncode == 0, p.stdout
    exp = expected(tiny_frames(N), np.ones((200, 300), np.uint8))
    rc, out, err, lines = run_with_pipe([str(master), "--threads", "2", "--batch", "2"], tmp_path)
    assert rc == 0 and not err, (out, err)
    got = {json.loads(l)["file-number"]: json.loads(l) for l in lines}
    assert sorted(got) == list(range(N))
    for i, (cc, _) in enumerate(exp):
        assert got[i]["num_strong_pixels"] == cc.num_strong_pixels
        assert got[i]["n_spots_total"] == len(cc.boxes)
        assert got[i]["file"] == str(master)
    rc, out, err, lines = run_with_pipe([str(master), "--dmin", "40"], tmp_path)
    assert rc == 0 and not err and len(lines) == N
    assert sum(json.loads(l)["num_strong_pixels"] for l in lines) < sum(cc.num_strong_pixels for cc, _ in exp)


def test_hdf5_rotation_sweep(tmp_path):
    """omega in the master switches the driver to 3D spot finding, as for the synthetic sweep."""
    master = tmp_path / "sweep_master.h5"
    p = subprocess.run([TOOL, "mkh5", "synth:tinysweep:8", str(master), "vds-files", "3"], capture_output=True, text=True)
    if "HDF5-enabled" in p.stdout:
        pytest.skip("built without HDF5")
    a = tmp_path / "a"
    b = tmp_path / "b"
    a.mkdir()
    b.mkdir()
    rc, out, err, _ = run_with_pipe([str(master), "--min-spot-size-3d", "2", "--writeout"], a)
    assert rc == 0 and not err, (out, err)
    rc2, out2, err2, _ = run_with_pipe(["synth:tinysweep:8", "--min-spot-size-3d", "2", "--writeout"], b)
    assert rc2 == 0
    assert (a / "3d_reflections.txt").read_text() == (b / "3d_reflections.txt").read_text()
    assert re.search(spots_match_regex, strip_ansi(out)).group(1) == re.search(spots_match_regex, strip_ansi(out2)).group(1)


def _h5stats(path):
    p = subprocess.run([TOOL, "h5stats", str(path), "dials/processing/group_0"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    out = {}
    for line in p.stdout.strip().split("\n"):
        f = line.split()
        rows, cols = int(f[1]), int(f[2])
        v = np.array([float(x) for x in f[3:]]).reshape(3, cols)   # min, max, mean per column
        out[f[0]] = (rows, v)
    return out


def test_save_h5_rotation(tmp_path):
    """results_ffs.h5 for a sweep (spotfinder.cc:1152-1262): centroids, ids, spot extents and the
    Kabsch-space variances, against the oracle's restatement; same summation order, so 1e-12."""
    from oracle import oracle as O
    if subprocess.run([TOOL, "h5support"], capture_output=True, text=True).stdout.strip() != "1":
        pytest.skip("built without HDF5")
    N = 10
    rc, out, err, _ = run_with_pipe(["synth:tinysweep:%d" % N, "--save-h5", "--min-spot-size-3d", "3", "--threads", "2",
                                     "--batch", "4"], tmp_path)
    assert rc == 0 and not err, (out, err)
    frames = tiny_frames(N, sweep=True)
    mask = np.ones((200, 300), np.uint8)
    exp = expected(frames, mask)
    slices = [(cc.k, cc.intensity) for cc, _ in exp]
    want = O.cc3d(slices, 300, 200, 3, 2.0)
    sig = O.cc3d_signals(slices, 300, 200, 3, 2.0)
    # the synthetic source's metadata (host/readers.cc): 0.3 m, 75 um pixels, beam centre at the middle
    # (the driver holds them as float32, spotfinder.cc:484-587; the products below are taken in double)
    f32 = lambda v: float(np.float32(v))
    geom = O.Geometry(f32(0.3) * 1000.0, 150.0, 100.0, f32(0.75e-4) * 1000.0, f32(0.75e-4) * 1000.0, f32(0.976), 0.0,
                      f32(0.1))
    sb, sm, depth = O.kabsch_variances(slices, 300, sig, want.reflections, geom)
    st = _h5stats(tmp_path / "results_ffs.h5")
    n = len(want.reflections)
    assert n > 5
    xyz = np.stack([want.reflections[c].astype(np.float64) for c in ("com_x", "com_y", "com_z")], axis=1)
    for name, col in (("xyzobs.px.value", xyz), ("sigma_b_variance", sb[:, None]), ("sigma_m_variance", sm[:, None]),
                      ("spot_extent_z", depth[:, None].astype(float)), ("id", np.zeros((n, 1)))):
        rows, v = st[name]
        assert rows == n, name
        ref = np.stack([col.min(0), col.max(0), col.mean(0)])
        np.testing.assert_allclose(v, ref, rtol=1e-8, atol=1e-300, err_msg=name)
    assert (sb > 0).all() and (sm[depth > 1] > 0).all()
    txt = strip_ansi(out)
    m = re.search(r"Estimated sigma_b \(degrees\): ([0-9.]+)", txt)
    assert m and abs(float(m.group(1)) - np.degrees(np.sqrt(sb.mean()))) < 1e-6
    deep = depth >= 5
    if deep.any():
        m = re.search(r"Estimated sigma_m \(degrees\): ([0-9.]+), calculated on (\d+) spots", txt)
        assert m and int(m.group(2)) == int(deep.sum())
        assert abs(float(m.group(1)) - np.degrees(np.sqrt(sm[deep].mean()))) < 1e-6
    assert "Successfully wrote 3D reflections to HDF5 file" in txt


def test_save_h5_stills(tmp_path):
    """2D: per-image centroids with one experiment id per image (spotfinder.cc:1265-1306)."""
    if subprocess.run([TOOL, "h5support"], capture_output=True, text=True).stdout.strip() != "1":
        pytest.skip("built without HDF5")
    N = 4
    rc, out, err, _ = run_with_pipe(["synth:tiny:%d" % N, "--save-h5", "--threads", "2"], tmp_path)
    assert rc == 0 and not err, (out, err)
    exp = expected(tiny_frames(N), np.ones((200, 300), np.uint8))
    xyz = np.concatenate([np.stack([r.reflections[c].astype(np.float64) for c in ("com_x", "com_y", "com_z")], axis=1)
                          for _, r in exp])
    ids = np.concatenate([np.full(len(r.reflections), i) for i, (_, r) in enumerate(exp)])
    st = _h5stats(tmp_path / "results_ffs.h5")
    rows, v = st["xyzobs.px.value"]
    assert rows == len(xyz) > 10
    np.testing.assert_allclose(v, np.stack([xyz.min(0), xyz.max(0), xyz.mean(0)]), rtol=1e-9)
    rows, v = st["id"]
    assert rows == len(ids) and v[1, 0] == N - 1 and abs(v[2, 0] - ids.mean()) < 1e-9
    assert "Successfully wrote %d 2D reflections to HDF5 file" % len(ids) in strip_ansi(out)
    assert "sigma_b_variance" not in st


def test_extended_algorithm_flag(tmp_path):
    """`-a dispersion_extended` (spotfinder.cc:338-342): per-frame counts from the oracle's extended mask."""
    from oracle import oracle as O
    N = 3
    frames = tiny_frames(N)
    rc, out, err, lines = run_with_pipe(["synth:tiny:%d" % N, "-a", "Dispersion_Extended", "--threads", "2"], tmp_path)
    assert rc == 0 and not err, (out, err)
    assert "Algorithm: Dispersion Extended" in out
    got = {json.loads(l)["file-number"]: json.loads(l) for l in lines}
    mask = np.ones((200, 300), np.uint8)
    for i, img in enumerate(frames):
        strong = O.dispersion_extended(img, mask)
        cc = O.cc2d(strong, img, 3)
        assert got[i]["num_strong_pixels"] == cc.num_strong_pixels == int(strong.sum())
        assert got[i]["n_spots_total"] == len(cc.boxes)


def test_dtype_exit_code_protocol(tmp_path):
    """With --strict-dtype the binary follows the reference's protocol: exit code = bit depth of the
    data when it does not match the executable (spotfinder.cc:468-476; service.py:503-507)."""
    p = subprocess.run([SPOTFINDER, "synth:jungfrau9m:1", "--strict-dtype"], capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 32 and "only accepts 16 bit" in p.stdout
    p = subprocess.run([os.path.join(BIN, "spotfinder32"), "synth:tiny:1", "--strict-dtype"], capture_output=True,
                       text=True, cwd=tmp_path)
    assert p.returncode == 16
    p = subprocess.run([SPOTFINDER, "--list-devices"], capture_output=True, text=True)
    assert p.returncode == 0 and re.match(r"0: .*gfx950", p.stdout.split("\n")[1])


def test_resolution_mask_flags(tmp_path):
    rc, out, err, lines = run_with_pipe(["synth:tiny:2", "--dmin", "40"], tmp_path)
    assert rc == 0 and not err
    rc2, out2, err2, lines2 = run_with_pipe(["synth:tiny:2"], tmp_path)
    a = [json.loads(l)["num_strong_pixels"] for l in sorted(lines)]
    b = [json.loads(l)["num_strong_pixels"] for l in sorted(lines2)]
    assert all(x <= y for x, y in zip(a, b)) and sum(a) < sum(b)


@pytest.mark.parametrize("transport", ["", "rccl", "peer"])
def test_several_device_contexts_share_the_frame_queue(tmp_path, transport):
    """--devices: one context and worker pool per listed GPU, all pulling from the one frame queue.  On a
    one-GPU box the list names the same GPU twice, which still makes two contexts, two worker pools and --
    for the rotation sweep -- the cross-context exchange of the strong-pixel lists into the 3D stack; with
    FFS_GATHER=rccl that exchange goes through RCCL send/recv (a one-rank communicator talking to itself)."""
    from oracle import oracle as O
    N = 10
    env = dict(os.environ)
    if transport:
        env["FFS_GATHER"] = transport
    argv = ["synth:tinysweep:%d" % N, "--devices", "0,0", "--threads", "4", "--batch", "2", "--writeout", "--min-spot-size-3d", "4"]
    r, w = os.pipe()
    proc = subprocess.Popen([SPOTFINDER, *argv, "--pipe_fd", str(w)], pass_fds=[w], cwd=tmp_path, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    os.close(w)
    out, err = proc.communicate(timeout=300)
    with os.fdopen(r) as f:
        lines = [json.loads(l) for l in f.read().split("\n") if l]
    assert proc.returncode == 0 and not err, err
    txt = strip_ansi(out)
    assert "GPUs:        0, 0" in txt
    if transport == "rccl":
        assert "exchange of rotation lists: rccl" in txt
    frames = tiny_frames(N, sweep=True)
    mask = np.ones((200, 300), np.uint8)
    exp = expected(frames, mask)
    # every frame exactly once, whatever GPU took it; JSON lines as with one device
    assert sorted(l["file-number"] for l in lines) == list(range(N))
    for l in lines:
        cc, _ = exp[l["file-number"]]
        assert l["num_strong_pixels"] == cc.num_strong_pixels and l["n_spots_total"] == len(cc.boxes)
    want = O.cc3d([(cc.k, cc.intensity) for cc, _ in exp], 300, 200, 4, 2.0)
    assert int(re.search(spots_match_regex, txt).group(1)) == want.n_calculated
    assert f"Found {len(want.reflections)} spots" in txt
    got = open(tmp_path / "3d_reflections.txt").read().strip().split("\n")
    assert len(got) == len(want.reflections) > 3
    for line, rr in zip(got, want.reflections):
        m = re.match(r"X: \[(\d+), (\d+)\] Y: \[(\d+), (\d+)\] Z: \[(\d+), (\d+)\]", line)
        assert [int(m.group(i)) for i in range(1, 7)] == [rr["x_min"], rr["x_max"], rr["y_min"], rr["y_max"], rr["z_min"], rr["z_max"]]


def test_unknown_device_in_list_is_refused(tmp_path):
    p = subprocess.run([SPOTFINDER, "synth:tiny:2", "--devices", "0,99"], capture_output=True, text=True, cwd=tmp_path)
    assert p.returncode == 1 and "device 99 does not exist" in p.stdout


def test_bench_single_process_leg_with_two_contexts_on_one_gpu():
    """bench.py --gpus 2 --single-process: one process, a context and a host thread per GPU, each inside the native submit /
    wait loop (ffs_bench_pipeline).  Rehearsed with both contexts on GPU 0 and the small workload."""
    import json
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0",
                        "--workload", "plumbing1k", "--batch", "4", "--steps", "6", "--warmup", "2", "--reps", "2"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["n_contexts_seen"] == 2 and d["value"] > 0
    assert d["config"]["spots_per_frame"] > 10 and d["config"]["strong_pixels_per_frame"] > 100


def _oracle_counts(frames, mask, max_valid=-1, min_count=2):
    from oracle import oracle as O
    p = O.DispParams()
    O.lib().ffs_oracle_default_disp_params(O.C.byref(p))
    p.min_count = min_count
    out = []
    for img in frames:
        strong = O.dispersion(img, mask, p)
        if max_valid >= 0:
            strong = strong & (img <= max_valid)               # kernels/thresholding.cu:208-215
        cc = O.cc2d(strong, img, 3)
        out.append((cc.num_strong_pixels, len(cc.boxes)))
    return out


def test_trusted_range_and_min_count_flags(tmp_path):
    """The reference driver hands the frame source's trusted maximum to every launch (spotfinder.cc:482,868,879) and its
    kernels refuse centre pixels above it (kernels/thresholding.cu:208-215); its launch wrapper defaults to min_count 3
    (spotfinder.cuh:18-20).  Here: `--max-valid trusted` (default) / `none` / `N` and `--min-count N`.  An Eiger-stream
    directory whose count-rate cut-off (800) lies far below the spots' peaks: counts against the oracle's mask ANDed with
    `img <= max_valid`, as tests/test_gpu_fuzz.py does for the library."""
    N = 4
    shm = tmp_path / "shm"
    assert subprocess.run([TOOL, "mkshm", "synth:tiny:%d" % N, str(shm)]).returncode == 0
    hdr = (shm / "start_1").read_text()
    assert '"countrate_correction_count_cutoff": 65535' in hdr
    (shm / "start_1").write_text(hdr.replace('"countrate_correction_count_cutoff": 65535', '"countrate_correction_count_cutoff": 800'))
    frames = tiny_frames(N)
    assert int((frames > 800).sum()) > 20
    mask = np.ones((200, 300), np.uint8)
    cases = [([], dict(max_valid=800)),                                     # default: trusted
             (["--max-valid", "trusted", "--cpu-decode"], dict(max_valid=800)),
             (["--max-valid", "none"], dict()),                             # the CPU baseline's behaviour
             (["--max-valid", "300", "--batch", "3"], dict(max_valid=300)),
             (["--min-count", "30"], dict(max_valid=800, min_count=30)),    # windows at the frame's edges hold fewer pixels
             (["--min-count", "3", "--max-valid", "none", "-a", "dispersion"], dict(min_count=3))]
    seen = set()
    for argv, want in cases:
        rc, out, err, lines = run_with_pipe([str(shm), "--threads", "2", *argv], tmp_path)
        assert rc == 0 and not err, (out, err)
        got = {json.loads(l)["file-number"]: json.loads(l) for l in lines}
        exp = _oracle_counts(frames, mask, **want)
        for i, (ns, nb) in enumerate(exp):
            assert (got[i]["num_strong_pixels"], got[i]["n_spots_total"]) == (ns, nb), (argv, i)
        seen.add(tuple(exp))
        assert ("Trusted range: centre pixels above" in out) == ("max_valid" in want)
    assert len(seen) >= 4                                                    # the flags changed the answers
    # a cut-off at or above the pixel type's maximum says nothing: no test (the synthetic source: 65535)
    rc, out, err, lines = run_with_pipe(["synth:tiny:2"], tmp_path)
    assert rc == 0 and "Trusted range" not in out
    r = subprocess.run([SPOTFINDER, "synth:tiny:2", "--min-count", "1"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 1 and "--min-count" in r.stdout


@pytest.mark.parametrize("algo", ["dispersion", "dispersion_extended"])
def test_validate_flag_cross_checks_every_image(tmp_path, algo):
    """`--validate` (spotfinder.cc:1012-1053: every image against the CPU baseline, "Compared: Match N px" / "Mismatch (N px from
    kernel)").  The CPU baseline is test infrastructure here; the flag runs every batch through a second context whose threshold
    stage gathers the window of EVERY valid pixel from memory (no streaming kernel) and compares the strong-pixel masks.
    The Match counts are the oracle's."""
    N = 5
    rc, out, err, lines = run_with_pipe(["synth:tiny:%d" % N, "--threads", "2", "--batch", "2", "--validate", "-a", algo], tmp_path)
    txt = strip_ansi(out)
    assert rc == 0 and not err, (out, err)
    m = dict((int(a), int(b)) for a, b in re.findall(r"Thread\s+\d+, Image\s+(\d+): Compared: Match (\d+) px", txt))
    assert sorted(m) == list(range(N)) and "Mismatch" not in txt
    assert f"Validation: 0 of {N} images differ" in txt
    from oracle import oracle as O
    frames = tiny_frames(N)
    mask = np.ones((200, 300), np.uint8)
    for i, img in enumerate(frames):
        strong = O.dispersion_extended(img, mask) if algo == "dispersion_extended" else O.dispersion(img, mask)
        assert m[i] == int(strong.sum())
    got = {json.loads(l)["file-number"]: json.loads(l) for l in lines}
    assert sorted(got) == list(range(N))


def _write_stream_dir(path, frames, mask=None):
    """An Eiger-stream directory (what `ffs_hosttool mkshm` writes) from arbitrary frames: start_1 header, start_5 pixel
    mask (int32, 0 = good) and one bitshuffle-LZ4 chunk per image."""
    from ffs_amd import bslz4
    os.makedirs(path)
    N, H, W = frames.shape
    hdr = {"nimages": N, "ntrigger": 1, "y_pixels_in_detector": H, "x_pixels_in_detector": W,
           "bit_depth_image": frames.dtype.itemsize * 8, "countrate_correction_count_cutoff": 65535 if frames.dtype == np.uint16 else 4294967295,
           "wavelength": 0.976, "detector_distance": 300.0, "y_pixel_size": 7.5e-05, "x_pixel_size": 7.5e-05,
           "beam_center_y": H / 2.0, "beam_center_x": W / 2.0}
    open(os.path.join(path, "start_1"), "w").write(json.dumps(hdr) + "\n")
    open(os.path.join(path, "start_4"), "w").write("\n")
    m = np.zeros((H, W), np.int32) if mask is None else (mask == 0).astype(np.int32)
    m.tofile(os.path.join(path, "start_5"))
    for i, f in enumerate(frames):
        open(os.path.join(path, "image_%06d_2" % i), "wb").write(bytes(bslz4.compress(f)))


@pytest.mark.parametrize("argv", [["--threads", "4", "--batch", "6"], ["--threads", "3", "--batch", "4", "--assemblies", "2"],
                                  ["--threads", "1", "--batch", "16"], ["--threads", "5", "--batch", "3", "--devices", "0,0"]])
def test_batches_assembled_by_several_readers_chunks_of_any_size(tmp_path, argv):
    """A GPU batch is filled by all reader threads of its GPU: image i goes to slot i mod B of batch i div B, slots are sized by
    the first chunk anybody reads (+ 2 %).  Here the first frames are empty (chunks of a few hundred bytes) and the later ones
    are noise and spots (chunks thousands of times larger): they take the overflow area behind the slots and, when that is
    full, the heap (their whole batch then goes up from there).  Every image must come out, in order, with the oracle's counts."""
    rng = np.random.default_rng(5)
    W, H, N = 300, 200, 23
    frames = np.zeros((N, H, W), np.uint16)
    for i in range(3, N):
        frames[i] = rng.poisson(2.0 if i % 3 else 40.0, (H, W))
        if i % 4 == 0:
            frames[i] = rng.integers(0, 4000, (H, W))                       # incompressible
        frames[i, 20 + i:23 + i, 50:53] += 500
    shm = tmp_path / "shm"
    _write_stream_dir(str(shm), frames)
    rc, out, err, lines = run_with_pipe([str(shm), *argv], tmp_path)
    assert rc == 0 and not err, (out, err)
    got = [json.loads(l) for l in lines]
    assert sorted(j["file-number"] for j in got) == list(range(N))
    exp = _oracle_counts(frames, np.ones((H, W), np.uint8))
    for j in got:
        assert (j["num_strong_pixels"], j["n_spots_total"]) == exp[j["file-number"]], j
    if "--devices" not in argv:
        assert [j["file-number"] for j in got] == list(range(N))              # one collector: results leave in frame order
    assert f"{N} images in" in out


def test_data_set_that_ends_early_reports_every_image_that_was_read(tmp_path):
    """A live stream directory whose header promises 11 images and holds 8: the readers give up after `--timeout`
    ("Timeout waiting for image 8", spotfinder.cc:776-787), and every image that WAS read still comes out -- the reference's
    workers finish the image they hold -- although the last GPU batch is only partly filled (batch 3: images 6, 7 of 6..8)."""
    rng = np.random.default_rng(9)
    W, H, N_have, N_said = 300, 200, 8, 11
    frames = rng.poisson(2.0, (N_have, H, W)).astype(np.uint16)
    for i in range(N_have):
        frames[i, 30 + i:33 + i, 60:63] += 400
    shm = tmp_path / "shm"
    _write_stream_dir(str(shm), frames)
    hdr = json.loads((shm / "start_1").read_text())
    hdr["nimages"] = N_said
    (shm / "start_1").write_text(json.dumps(hdr) + "\n")
    for argv in (["--threads", "3", "--batch", "3"], ["--threads", "2", "--batch", "4", "--cpu-decode"]):
        rc, out, err, lines = run_with_pipe([str(shm), "--timeout", "1.5", *argv], tmp_path)
        assert rc == 0 and not err, (out, err)
        assert re.search(r"Timeout waiting for image (8|9|10)\b", out)   # (whichever reader's patience runs out first says so)
        got = [json.loads(l) for l in lines]
        assert [j["file-number"] for j in got] == list(range(N_have))
        exp = _oracle_counts(frames, np.ones((H, W), np.uint8))
        for j in got:
            assert (j["num_strong_pixels"], j["n_spots_total"]) == exp[j["file-number"]]
        assert f"{N_have} images in" in out


def test_interrupt_stops_the_readers_and_reports_what_was_read(tmp_path):
    """SIGINT (spotfinder.cc:43-54, "Running interrupted by user request"): the readers stop, what has been read is processed and
    reported, the summary line counts exactly the JSON lines that went out, exit code 0."""
    import signal
    import time
    r, w = os.pipe()
    proc = subprocess.Popen([SPOTFINDER, "synth:tiny:200000", "--threads", "2", "--pipe_fd", str(w)], pass_fds=[w], cwd=tmp_path,
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    os.close(w)
    lines = []
    import threading

    def drain():
        with os.fdopen(r) as f:
            for ln in f:
                lines.append(ln)
    th = threading.Thread(target=drain, daemon=True)
    th.start()
    time.sleep(1.5)
    proc.send_signal(signal.SIGINT)
    out, err = proc.communicate(timeout=60)
    th.join(10)
    assert proc.returncode == 0 and not err, (out[-500:], err)
    assert "Running interrupted by user request" in out
    m = re.search(r"(\d+) images in", strip_ansi(out))
    assert m and 0 < int(m.group(1)) < 200000
    nums = [json.loads(l)["file-number"] for l in lines if l.strip()]
    assert len(nums) == int(m.group(1)) and nums == list(range(len(nums)))      # a contiguous prefix, in order, nothing lost


def test_interrupt_wakes_readers_parked_for_a_free_assembly(tmp_path):
    """More readers than slots (20 threads, K = 3 assemblies of 2 images): after the six images the live stream directory holds,
    six readers poll for images that never come and the others are parked waiting for a free assembly -- which only a completed
    batch frees, and after SIGINT none completes.  One signal must end the run: the parked readers are told (the signal handler
    itself can only set a flag), the six images that were read are reported, exit code 0.  Reference: its workers poll the stop
    flag between images, spotfinder.cc:770-790."""
    import signal
    import time
    rng = np.random.default_rng(11)
    W, H, N_have, N_said = 300, 200, 6, 100
    frames = rng.poisson(2.0, (N_have, H, W)).astype(np.uint16)
    for i in range(N_have):
        frames[i, 30 + i:33 + i, 60:63] += 400
    shm = tmp_path / "shm"
    _write_stream_dir(str(shm), frames)
    hdr = json.loads((shm / "start_1").read_text())
    hdr["nimages"] = N_said
    (shm / "start_1").write_text(json.dumps(hdr) + "\n")
    r, w = os.pipe()
    proc = subprocess.Popen([SPOTFINDER, str(shm), "--cpu-decode", "--threads", "20", "--batch", "2", "--timeout", "120", "--pipe_fd", str(w)],
                            pass_fds=[w], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    os.close(w)
    time.sleep(3.0)
    t0 = time.time()
    proc.send_signal(signal.SIGINT)
    out, err = proc.communicate(timeout=60)          # (a hang here is the bug: TimeoutExpired)
    assert time.time() - t0 < 20
    with os.fdopen(r) as f:
        lines = [l for l in f.read().split("\n") if l]
    assert proc.returncode == 0 and not err, (out[-500:], err)
    assert "Running interrupted by user request" in out
    assert [json.loads(l)["file-number"] for l in lines] == list(range(N_have))
    assert f"{N_have} images in" in strip_ansi(out)


def test_clean_exit_flag_keeps_the_full_teardown(tmp_path):
    """By default the process leaves as soon as its last result is out (the service starts one per request); `--clean-exit`
    destroys streams and contexts and lets the runtime tear itself down.  Same lines, same exit code, nothing on stderr either way."""
    outs = []
    for extra in ([], ["--clean-exit"]):
        rc, out, err, lines = run_with_pipe(["synth:tiny:9", "--threads", "2", "--batch", "4", *extra], tmp_path)
        assert rc == 0 and not err, (out[-300:], err)
        outs.append([json.loads(l) for l in lines])
        assert "9 images in" in strip_ansi(out)
    assert outs[0] == outs[1] and [j["file-number"] for j in outs[0]] == list(range(9))


def test_gather_rccl_gives_the_same_spot_centres_as_the_host_read(tmp_path):
    """`--gpus N --gather rccl --output-for-index`: the spot centres of every round of batches (one per GPU) come back through
    ffs_multi_gather_rows (counts by ncclAllGather, rows by ncclSend / ncclRecv to the first GPU) instead of being read from each
    context's host arrays (`--gather host`, the default).  Two contexts on this one GPU (a one-rank communicator, self send / recv);
    23 images in batches of 4: five full rounds and a last one that cannot fill up (served from the host arrays).  Same JSON lines."""
    lines = {}
    for how in ("host", "rccl"):
        rc, out, err, ls = run_with_pipe(["synth:tiny:23", "--threads", "4", "--batch", "2", "--devices", "0,0", "--output-for-index", "--gather", how], tmp_path)
        assert rc == 0 and not err, (out[-400:], err)
        lines[how] = sorted((json.loads(l) for l in ls), key=lambda j: j["file-number"])
        if how == "rccl":
            m = re.search(r"Spot lists: (\d+) rounds of 2 batches gathered over RCCL", out)
            if "exchange of rotation lists" in out and not m:
                pytest.skip("no RCCL on this machine")
            assert m and int(m.group(1)) >= 5, out[-600:]
    assert [j["file-number"] for j in lines["host"]] == list(range(23))
    assert lines["host"] == lines["rccl"]
    assert sum(len(j["spot_centers"]) for j in lines["host"]) > 0
