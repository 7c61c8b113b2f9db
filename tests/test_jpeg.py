"""The CLI's baseline JPEG reader (opticalflowhs_amd/csrc/host/jpeg_baseline.hpp) against PIL's decoder
(libjpeg-turbo), bit for bit: the reference's own output pictures (data under tests/golden) and synthetic
files over sizes, chroma samplings, qualities, grayscale and restart intervals.  CPU-only."""
import io
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

Image = pytest.importorskip("PIL.Image")


@pytest.fixture(scope="module")
def jpeg2ppm(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("jpeg") / "jpeg2ppm")
    src = os.path.join(ROOT, "opticalflowhs_amd", "csrc", "host", "jpeg2ppm.cpp")
    r = subprocess.run([gxx, "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def ours(exe, jpg_path, tmp):
    out = os.path.join(tmp, "o.ppm")
    r = subprocess.run([exe, jpg_path, out], capture_output=True, text=True)
    if r.returncode != 0:
        return None, r.stderr
    with open(out, "rb") as f:
        magic = f.readline().strip()
        w, h = [int(t) for t in f.readline().split()]
        f.readline()
        data = np.frombuffer(f.read(), np.uint8)
    return (data.reshape(h, w, 3) if magic == b"P6" else data.reshape(h, w)), ""


def test_reference_pictures_decode_like_pil(jpeg2ppm, tmp_path):
    for name in ("ref_city_cv_out.jpg", "ref_bunny_cv_out.jpg", "ref_city_cl_out.jpg", "ref_bunny_cl_out.jpg"):
        path = os.path.join(GOLDEN, name)
        a, err = ours(jpeg2ppm, path, str(tmp_path))
        assert a is not None, err
        assert np.array_equal(a, np.asarray(Image.open(path).convert("RGB"))), name


def test_synthetic_files_decode_like_pil(jpeg2ppm, tmp_path):
    rng = np.random.default_rng(5)
    n = 0
    for case in range(80):
        W = int(rng.integers(1, 90)) if case % 3 else int(rng.integers(1, 20))
        H = int(rng.integers(1, 70)) if case % 4 else int(rng.integers(1, 18))
        gray = case % 7 == 0
        kind = case % 3
        if kind == 0:      # noise: exercises long Huffman codes and clamping
            arr = rng.integers(0, 256, size=(H, W) if gray else (H, W, 3), dtype=np.uint8)
        elif kind == 1:    # smooth gradients: DC-only blocks, zero runs, EOB
            yy, xx = np.mgrid[0:H, 0:W]
            base = (xx * 3 + yy * 2) % 256
            arr = base.astype(np.uint8) if gray else np.stack([base, (base * 2) % 256, 255 - base], axis=2).astype(np.uint8)
        else:              # flat colour with one bright rectangle
            arr = np.full((H, W) if gray else (H, W, 3), 40, np.uint8)
            arr[H // 3:H // 3 + max(1, H // 4), W // 3:W // 3 + max(1, W // 4)] = 230
        kw = dict(quality=int(rng.choice([20, 50, 75, 90, 95, 100])))
        if not gray:
            kw["subsampling"] = int(rng.choice([0, 1, 2]))   # 4:4:4, 4:2:2, 4:2:0
        if case % 5 == 0:
            kw["restart_marker_blocks"] = int(rng.integers(1, 6))
        buf = io.BytesIO()
        try:
            Image.fromarray(arr).save(buf, format="JPEG", **kw)
        except (TypeError, OSError):  # an older Pillow without restart_marker_blocks
            kw.pop("restart_marker_blocks", None)
            buf = io.BytesIO()
            Image.fromarray(arr).save(buf, format="JPEG", **kw)
        path = str(tmp_path / "c.jpg")
        with open(path, "wb") as f:
            f.write(buf.getvalue())
        a, err = ours(jpeg2ppm, path, str(tmp_path))
        assert a is not None, (case, W, H, kw, err)
        ref = np.asarray(Image.open(path))
        assert a.shape == ref.shape and np.array_equal(a, ref), (case, W, H, gray, kw, int(np.abs(a.astype(int) - ref.astype(int)).max()))
        n += 1
    assert n == 80


def test_unsupported_and_broken_files_are_refused(jpeg2ppm, tmp_path):
    arr = np.random.default_rng(1).integers(0, 256, size=(40, 50, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="JPEG", progressive=True)
    p = str(tmp_path / "prog.jpg")
    open(p, "wb").write(buf.getvalue())
    a, err = ours(jpeg2ppm, p, str(tmp_path))
    assert a is None and "progressive" in err
    buf = io.BytesIO()
    Image.fromarray(arr).save(buf, format="JPEG")
    data = buf.getvalue()
    for cut in (3, 20, len(data) // 3):
        p = str(tmp_path / "cut.jpg")
        open(p, "wb").write(data[:cut])
        a, err = ours(jpeg2ppm, p, str(tmp_path))
        assert a is None or a.shape == (40, 50, 3)   # refused, or decoded with the missing part zero-filled: never a crash
    p = str(tmp_path / "not.jpg")
    open(p, "wb").write(b"P6\n1 1\n255\n\x00\x00\x00")
    a, err = ours(jpeg2ppm, p, str(tmp_path))
    assert a is None and "not a JPEG" in err


def test_reference_inputs_give_the_committed_gray_planes(jpeg2ppm, oracle, tmp_path):
    """The reference's own input pictures through the CLI's reader and BGR2GRAY = the gray planes every
    other test works on (those were made with PIL's decoder)."""
    import refpics
    for name in ("city", "bunny"):
        for k in (1, 2):
            rgb, err = ours(jpeg2ppm, os.path.join(GOLDEN, "ref_%s_%d.jpg" % (name, k)), str(tmp_path))
            assert rgb is not None, err
            gray = oracle.bgr2gray(np.ascontiguousarray(rgb[:, :, ::-1]))
            assert np.array_equal(gray, refpics.read_pgm(os.path.join(GOLDEN, "%s_%d_gray.pgm" % (name, k)))), (name, k)


@pytest.fixture(scope="module")
def ppm2jpeg(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("jpegw") / "ppm2jpeg")
    src = os.path.join(ROOT, "opticalflowhs_amd", "csrc", "host", "ppm2jpeg.cpp")
    r = subprocess.run([gxx, "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", exe, src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def encode_ours(exe, arr, tmp, quality=95):
    src, dst = os.path.join(tmp, "e.ppm"), os.path.join(tmp, "e.jpg")
    with open(src, "wb") as f:
        f.write(b"P%d\n%d %d\n255\n" % (6 if arr.ndim == 3 else 5, arr.shape[1], arr.shape[0]))
        f.write(np.ascontiguousarray(arr).tobytes())
    subprocess.check_call([exe, src, dst, str(quality)])
    return open(dst, "rb").read()


def test_writer_is_byte_identical_to_libjpeg(ppm2jpeg, tmp_path):
    """The CLI's JPEG writer (host/jpeg_encode.hpp: what cvSaveImage does) against PIL's encoder with the
    same settings (quality, 4:2:0, standard tables): the same bytes, for ragged sizes, gray and colour."""
    rng = np.random.default_rng(9)
    for case in range(80):
        W, H = int(rng.integers(1, 80)), int(rng.integers(1, 60))
        gray = case % 5 == 0
        shape = (H, W) if gray else (H, W, 3)
        arr = rng.integers(0, 256, size=shape, dtype=np.uint8) if case % 3 else np.full(shape, 17 * (case % 15), np.uint8)
        q = int(rng.choice([95, 95, 75, 50, 30, 100, 10]))
        buf = io.BytesIO()
        Image.fromarray(arr).save(buf, format="JPEG", quality=q, **({} if gray else {"subsampling": 2}))
        assert encode_ours(ppm2jpeg, arr, str(tmp_path), q) == buf.getvalue(), (case, W, H, gray, q)


def test_writer_reproduces_the_reference_files(ppm2jpeg, oracle, tmp_path):
    """Oracle flow -> drawing rule -> this writer = the reference's output FILES, byte for byte (all four)."""
    import refpics
    for name in ("city", "bunny"):
        A0, B0 = refpics.gray_pair(name)
        u, v = oracle.calc_optical_flow_hs(oracle.box_blur3(A0), oracle.box_blur3(B0), refpics.LAMBDA, refpics.ITERATIONS,
                                           epsilon=refpics.EPSILON, term_type=3)
        assert encode_ours(ppm2jpeg, refpics.render(u, v), str(tmp_path)) == open(os.path.join(GOLDEN, "ref_%s_cv_out.jpg" % name), "rb").read()
        u, v = oracle.classic_flow(A0, B0, refpics.ALPHA, refpics.ITERATIONS, update_v=False)
        assert encode_ours(ppm2jpeg, refpics.render(u, v, "cl"), str(tmp_path)) == open(os.path.join(GOLDEN, "ref_%s_cl_out.jpg" % name), "rb").read()
