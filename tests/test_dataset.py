"""Row f-4 of SURVEY.md section 8: ingest.  The mirrored EurocDataset / DataProvider (ocean-perception_amd/host/
dataset.hpp) with its own PNG / PNM reader, driven like the reference's Sequence demo
(test/stereo_matching/patchmatch_gpu_test.cpp:95-138)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT, assert_same, small_pair

PKG = os.path.join(ROOT, "ocean-perception_amd")
LIBDIR = os.path.join(PKG, "lib")


def write_png(path, img, filters=(0, 1, 2, 3, 4)):
    """Minimal PNG encoder (8-bit gray or RGB[A]); scan lines cycle through all five filter types so the
    reader's unfiltering is exercised."""
    img = np.ascontiguousarray(img)
    rows, cols = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    flat = img.reshape(rows, cols * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(cols * ch, np.int32)
    for y in range(rows):
        f = filters[y % len(filters)]
        cur = flat[y]
        a = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        c = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if f == 0:
            line = cur
        elif f == 1:
            line = cur - a
        elif f == 2:
            line = cur - prev
        elif f == 3:
            line = cur - (a + prev) // 2
        else:
            p = a + prev - c
            pa, pb, pc = np.abs(p - a), np.abs(p - prev), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            line = cur - pred
        raw.append(f)
        raw += (line & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2  # two IDAT chunks: the reader must concatenate them
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", cols, rows, 8, ctype, 0, 0, 0)) +
                chunk(b"tEXt", b"Comment\x00synthetic") + chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) +
                chunk(b"IEND", b""))


@pytest.fixture(scope="module")
def dataset_exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("cppds") / "dataset_main"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(PKG, "host"), os.path.join(ROOT, "tests", "cpp", "dataset_main.cpp"), "-L" + LIBDIR,
           "-lvehicle_pm_gpu", "-Wl,-rpath," + LIBDIR, "-pthread", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)


def _read(exe, path, tmp_path):
    out = os.path.join(tmp_path, "img.raw")
    r = subprocess.run([exe, "read", path, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rows, cols, ch = map(int, r.stdout.split())
    data = np.fromfile(out, np.uint8)
    gray = np.fromfile(out + ".gray", np.uint8).reshape(rows, cols) if ch == 3 else None
    return (data.reshape(rows, cols) if ch == 1 else data.reshape(rows, cols, 3)), gray


def test_image_reader_png_and_pnm(dataset_exe, tmp_path):
    """No GPU involved: the decoder and the 8-bit BGR2GRAY."""
    rng = np.random.default_rng(0)
    g = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    rgb = rng.integers(0, 256, (21, 34, 3), dtype=np.uint8)
    rgba = np.concatenate([rgb, rng.integers(0, 256, (21, 34, 1), dtype=np.uint8)], -1)
    want_gray = ((rgb[..., 2].astype(np.int64) * 1868 + rgb[..., 1].astype(np.int64) * 9617 +
                  rgb[..., 0].astype(np.int64) * 4899 + 8192) >> 14).astype(np.uint8)
    p = os.path.join(tmp_path, "g.png")
    write_png(p, g)
    got, _ = _read(dataset_exe, p, tmp_path)
    assert np.array_equal(got, g)
    for name, img in (("rgb.png", rgb), ("rgba.png", rgba)):
        p = os.path.join(tmp_path, name)
        write_png(p, img)
        got, gray = _read(dataset_exe, p, tmp_path)
        assert np.array_equal(got, rgb[..., ::-1]), "file order RGB -> BGR, alpha dropped"
        assert np.array_equal(gray, want_gray)
    p = os.path.join(tmp_path, "g.pgm")
    with open(p, "wb") as f:
        f.write(b"P5\n# comment\n%d %d\n255\n" % (g.shape[1], g.shape[0]) + g.tobytes())
    got, _ = _read(dataset_exe, p, tmp_path)
    assert np.array_equal(got, g)
    p = os.path.join(tmp_path, "c.ppm")
    with open(p, "wb") as f:
        f.write(b"P6 %d %d 255\n" % (rgb.shape[1], rgb.shape[0]) + rgb.tobytes())
    got, gray = _read(dataset_exe, p, tmp_path)
    assert np.array_equal(got, rgb[..., ::-1]) and np.array_equal(gray, want_gray)
    # corrupt CRC and unsupported files fail loudly
    p = os.path.join(tmp_path, "bad.png")
    data = bytearray(open(os.path.join(tmp_path, "g.png"), "rb").read())
    data[40] ^= 0xff
    open(p, "wb").write(data)
    r = subprocess.run([dataset_exe, "read", p, os.path.join(tmp_path, "x")], capture_output=True, text=True)
    assert r.returncode == 10 and "PNG" in r.stdout


def _pil():
    return pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("size,subsampling,quality", [((64, 48), 2, 90), ((53, 37), 2, 75), ((53, 37), 1, 85),
                                                      ((40, 30), 0, 95), ((17, 9), 2, 60), ((129, 65), 2, 30)])
def test_jpeg_decoder_matches_libjpeg_turbo(dataset_exe, tmp_path, size, subsampling, quality):
    """The own baseline JPEG decoder against Pillow's (libjpeg-turbo, accurate integer IDCT, fancy upsampling):
    luma plane (IMREAD_GRAYSCALE) and BGR output, bit for bit.  No GPU involved."""
    Image = _pil()
    w, h = size
    rng = np.random.default_rng(w * h + subsampling)
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([127 + 100 * np.sin(xx / 7.0) * np.cos(yy / 5.0), 127 + 90 * np.cos(xx / 3.0 + yy / 11.0),
                    60 + 2.0 * xx + rng.normal(0, 12, (h, w))], -1).clip(0, 255).astype(np.uint8)
    p = os.path.join(tmp_path, "c.jpg")
    Image.fromarray(rgb).save(p, quality=quality, subsampling=subsampling)
    want_rgb = np.asarray(Image.open(p).convert("RGB"))
    im = Image.open(p)
    im.draft("L", im.size)  # the decoder's own grayscale output = the luma plane
    want_y = np.asarray(im)
    assert want_y.shape == (h, w)
    got, _ = _read(dataset_exe, p, tmp_path)
    assert got.shape == (h, w, 3)
    assert np.array_equal(got, want_rgb[..., ::-1]), np.abs(got.astype(int) - want_rgb[..., ::-1].astype(int)).max()
    out = os.path.join(tmp_path, "y.raw")
    r = subprocess.run([dataset_exe, "readgray", p, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(out, np.uint8).reshape(h, w), want_y)
    # a grayscale JPEG
    pg = os.path.join(tmp_path, "g.jpg")
    Image.fromarray(rgb[..., 1]).save(pg, quality=quality)
    got, _ = _read(dataset_exe, pg, tmp_path)
    assert np.array_equal(got, np.asarray(Image.open(pg)))


def test_jpeg_decoder_on_the_reference_test_images(dataset_exe, tmp_path):
    """test/resources/caddy_32_{left,right}_small.jpg of the reference (fixtures: tests/golden/): the inputs of its
    stereo tests are JPEGs read with cv::imread(..., GRAYSCALE)."""
    Image = _pil()
    for name in ("caddy_32_left_small.jpg", "caddy_32_right_small.jpg"):
        p = os.path.join(ROOT, "tests", "golden", name)
        im = Image.open(p)
        im.draft("L", im.size)
        want = np.asarray(im)
        out = os.path.join(tmp_path, "y.raw")
        r = subprocess.run([dataset_exe, "readgray", p, out], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(out, np.uint8).reshape(want.shape), want)
        got, _ = _read(dataset_exe, p, tmp_path)
        assert np.array_equal(got, np.asarray(Image.open(p).convert("RGB"))[..., ::-1])


def test_jpeg_decoder_rejects_malformed_input_under_asan(tmp_path):
    """Every truncation, 3000 random mutations and crafted header fields (table selectors > 3, scan component
    counts 0 / too large, a second frame header, short table segments, DC categories >= 16) of two small JPEGs go
    through the decoder built with AddressSanitizer + UBSan (CPU build): each input is decoded or rejected with an
    exception, never a sanitizer report.  The reference uses cv::imdecode (lcm_util/decode_image.cpp:11-32)."""
    Image = _pil()
    exe = os.path.join(tmp_path, "jpeg_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host"),
           os.path.join(ROOT, "tests", "cpp", "jpeg_fuzz_main.cpp"), os.path.join(PKG, "host", "jpeg.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:24, 0:40]
    rgb = np.stack([127 + 100 * np.sin(xx / 5.0), 127 + 90 * np.cos(yy / 3.0), rng.integers(0, 255, (24, 40))],
                   -1).clip(0, 255).astype(np.uint8)
    for i, kw in enumerate((dict(quality=80, subsampling=2), dict(quality=60, subsampling=0))):
        p = os.path.join(tmp_path, f"s{i}.jpg")
        Image.fromarray(rgb).save(p, **kw)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
        r = subprocess.run([exe, p, "3000"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
        ok, bad = map(int, r.stdout.split()[1:3])
        assert ok > 100 and bad > 500, r.stdout  # both paths were exercised


@pytest.mark.gpu
def test_euroc_playback_through_the_pipelined_matcher(dataset_exe, tmp_path, oracle, synth):
    rows, cols, n = 64, 112, 5
    root = os.path.join(tmp_path, "euroc")
    pairs = []
    for cam in ("cam0", "cam1"):
        os.makedirs(os.path.join(root, "mav0", cam, "data"))
    stamps = [1403636579763555584 + i * 50000000 for i in range(n)]
    for cam in ("cam0", "cam1"):
        with open(os.path.join(root, "mav0", cam, "data.csv"), "w") as f:
            f.write("#timestamp [ns],filename\n")
            for t in stamps:
                f.write(f"{t},{t}.png\n")
    for i, t in enumerate(stamps):
        l, r, _, _, _ = small_pair(synth, 200 + i, rows, cols, n_points=25, dilate_factor=2)
        pairs.append((l, r))
        write_png(os.path.join(root, "mav0", "cam0", "data", f"{t}.png"), l)
        write_png(os.path.join(root, "mav0", "cam1", "data", f"{t}.png"), r)
    out_dir = os.path.join(tmp_path, "out")
    os.makedirs(out_dir)
    res = subprocess.run([dataset_exe, "play", root, out_dir, "2"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert f"played {n} pairs, collected {n}" in res.stdout
    for i, (l, r) in enumerate(pairs):
        got = np.fromfile(os.path.join(out_dir, f"disp_{i}.f32"), np.float32).reshape(rows, cols)
        # Match() seeds itself (SparseInit on both views), as the reference does
        sl = oracle.sparse_init(l, r, 4)
        sr = oracle.sparse_init(r[:, ::-1], l[:, ::-1], 4)[:, ::-1]
        el, _ = oracle.match(oracle.default_params(0, patch=5, n_iters=2, nthreads=8), l, r, sl, sr)
        assert_same(got, el, f"pair {i}")
