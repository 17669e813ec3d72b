#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: regenerate tests/golden/ from the reference's own scene layer.

Runs only where /root/reference exists (this container).  It executes oracle/_ref/ref_dump --
a tool linked against the reference's CRT* sources compiled in place (oracle/Makefile) -- and writes

  tests/golden/dragon_scene_layer.json   known answers of the reference scene layer for its one shipped
                                         scene: sizeof()s, parse results, CRTMesh::calculateVertexNormals
                                         output (R/CRTMesh.cpp:66-94), CRTCamera operation sequences
                                         (R/CRTCamera.cpp:9-130), CRTVector*CRTMatrix (R/CRTMatrix.cpp:26-38)
  tests/golden/texture_known_answers.json  getColor(u,v) of the reference's four texture classes (R/CRTTexture*.cpp) on a grid,
                                         the bitmap one reading tests/golden/tex7x5.ppm (written here, seeded)
  tests/golden/dragon.crtscene           the same scene DATA re-serialised from the reference's parsed
                                         values (float32 printed with 9 significant digits round-trips
                                         exactly) so the GPU box, which has no /root/reference, can load it.
"""
import json, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SCENE = "/root/reference/DirectX-RayTracer/DirectX-RayTracer/Scenes/Dragon.crtscene"
MATERIAL_NAMES = {1: "diffuse", 2: "reflective", 3: "refractive", 4: "constant"}


def fmt(x):
    return repr(float(x)) if float(x) != int(x) else str(int(x))


def main():
    if not os.path.exists(REF_SCENE):
        print("reference tree absent; nothing to do")
        return 0
    subprocess.check_call(["make", "-C", HERE, "_ref/ref_dump"])
    tmp = os.path.join(HERE, "_ref", "dragon_ref.json")
    subprocess.check_call([os.path.join(HERE, "_ref", "ref_dump"), REF_SCENE, tmp])
    d = json.load(open(tmp))
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)

    # 1) re-serialised scene data
    scene = {
        "settings": {"background_color": d["settings"]["background_color"],
                     "image_settings": {"width": d["settings"]["width"], "height": d["settings"]["height"]}},
        "camera": {"matrix": d["camera"]["matrix"], "position": d["camera"]["position"]},
        "lights": [{"intensity": l["intensity"], "position": l["position"]} for l in d["lights"]],
        "materials": [{"type": MATERIAL_NAMES[m["type"]], "albedo": m["albedo"],
                       "smooth_shading": bool(m["smooth_shading"])} for m in d["materials"]],
        "objects": [{"material_index": m["material_index"],
                     "vertices": [c for v in m["vertices"] for c in v],
                     "triangles": m["indices"]} for m in d["meshes"]],
    }
    with open(os.path.join(gold, "dragon.crtscene"), "w") as f:
        json.dump(scene, f, separators=(",", ":"))

    # 2) known answers (drop the bulky raw geometry: it lives in the scene file above)
    for m in d["meshes"]:
        del m["vertices"], m["indices"]
    with open(os.path.join(gold, "dragon_scene_layer.json"), "w") as f:
        json.dump(d, f, separators=(",", ":"))
    # 3) texture classes: a small binary PPM the reference's vendored stb_image can decode, and getColor known answers
    import numpy as np
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    with open(os.path.join(gold, "tex7x5.ppm"), "wb") as f:
        f.write(b"P6\n7 5\n255\n" + img.tobytes())
    ttmp = os.path.join(HERE, "_ref", "textures_ref.json")
    subprocess.check_call([os.path.join(HERE, "_ref", "ref_dump"), "--textures", os.path.join(gold, "tex7x5.ppm"), ttmp])
    with open(os.path.join(gold, "texture_known_answers.json"), "w") as f:
        json.dump(json.load(open(ttmp)), f, separators=(",", ":"))
    # 4) bitmap textures in the other formats the reference's stb_image reads and this repo decodes itself (csrc/image_decode.cpp):
    #    seeded images written here (PNG: every colour type, 1 / 4 / 8 / 16 bits, all five filters, stored / fixed / dynamic deflate
    #    blocks, Adam7; BMP: 24 / 32 bit, palette, top-down; TGA: raw / run-length, colour / grey; JPEG: baseline / progressive,
    #    4:4:4 / 4:2:2 / 4:2:0 / 4:1:1, grey, CMYK, restart intervals, one-texel edges; GIF; Radiance HDR; PSD; Softimage PIC), answers by CRTTextureBitmap
    answers = {}
    for name, data in bitmap_fixtures().items():
        path = os.path.join(gold, name)
        with open(path, "wb") as f:
            f.write(data)
        btmp = os.path.join(HERE, "_ref", "bitmap_ref.json")
        subprocess.check_call([os.path.join(HERE, "_ref", "ref_dump"), "--bitmap", path, btmp])
        # stored compactly: [iu, iv, R, G, B] with u = fl(iu * 0.05f + (iu % 3) * 0.003f), v = fl(iv * 0.05f + (iv % 4) * 0.002f) and
        # the colour the reference returned = fl(R / 255.0f) etc. exactly (checked here)
        import numpy as np
        rows = []
        k = 0
        for iu in range(21):
            for iv in range(21):
                u, v, r, g, b = json.load(open(btmp))[k] if k == 0 else ref[k]
                if k == 0:
                    ref = json.load(open(btmp))
                k += 1
                f32 = np.float32
                assert f32(u) == f32(f32(iu) * f32(0.05) + f32(iu % 3) * f32(0.003)) and f32(v) == f32(f32(iv) * f32(0.05) + f32(iv % 4) * f32(0.002)), (name, iu, iv)
                if name in ONE_CHANNEL:
                    # R/CRTTextureBitmap.cpp:27-31 reads buffer[index + 1] for green whatever the channel count: on the LAST texel of a
                    # one-channel image that is one byte past stbi_load's buffer -- undefined, whatever the heap holds -- so that
                    # sample is no known answer
                    wI, hI = ONE_CHANNEL[name]
                    uu, vv = min(max(f32(u), f32(0)), f32(1)), min(max(f32(v), f32(0)), f32(1))
                    if int((f32(1) - vv) * f32(hI - 1)) == hI - 1 and int(uu * f32(wI - 1)) == wI - 1:
                        continue
                rgb = [int(round(c * 255.0)) for c in (r, g, b)]
                assert all(f32(c) == f32(f32(q) / f32(255.0)) for c, q in zip((r, g, b), rgb)), (name, iu, iv, r, g, b)
                rows.append([iu, iv] + rgb)
        answers[name] = rows
    with open(os.path.join(gold, "bitmap_known_answers.json"), "w") as f:
        json.dump(answers, f, separators=(",", ":"))
    print("wrote", gold)
    return 0


def _png(w, h, depth, color, rows, level=9, strategy=None, interlace=False, palette=None, trns=None, split_idat=1):
    """rows: h byte strings of raw (unfiltered) scanline bytes.  Scanline y is filtered with type y % 5."""
    import struct
    import zlib
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    bpp = max(1, ch * depth // 8)

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if pa <= pb and pa <= pc else (b if pb <= pc else c)

    def filtered(lines):
        out = bytearray()
        prev = None
        for y, cur in enumerate(lines):
            t = y % 5
            out.append(t)
            for i, v in enumerate(cur):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i] if prev is not None else 0
                c = prev[i - bpp] if (prev is not None and i >= bpp) else 0
                pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][t]
                out.append((v - pred) & 255)
            prev = cur
        return bytes(out)

    if not interlace:
        raw = filtered(rows)
    else:
        x0, y0, dx, dy = [0, 4, 0, 2, 0, 1, 0], [0, 0, 4, 0, 2, 0, 1], [8, 8, 4, 4, 2, 2, 1], [8, 8, 8, 4, 4, 2, 2]
        px = ch * depth // 8
        raw = b""
        for p in range(7):
            lines = []
            for y in range(y0[p], h, dy[p]):
                if depth >= 8:
                    line = b"".join(rows[y][x * px:(x + 1) * px] for x in range(x0[p], w, dx[p]))
                else:  # one channel of 1 / 2 / 4 bits: unpack the row's samples, take the pass's columns, pack them again
                    vals = [(rows[y][(x * depth) >> 3] >> (8 - depth - ((x * depth) & 7))) & ((1 << depth) - 1) for x in range(x0[p], w, dx[p])]
                    acc = bytearray((len(vals) * depth + 7) // 8)
                    for i, v in enumerate(vals):
                        acc[(i * depth) >> 3] |= v << (8 - depth - ((i * depth) & 7))
                    line = bytes(acc)
                if line:
                    lines.append(line)
            if lines:
                raw += filtered(lines)
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY if strategy is None else strategy)
    z = co.compress(raw) + co.flush()

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    out += chunk(b"tEXt", b"Comment\0seeded test image")
    n = max(1, len(z) // split_idat)
    for i in range(0, len(z), n):
        out += chunk(b"IDAT", z[i:i + n])
    return out + chunk(b"IEND", b"")


ONE_CHANNEL = {"tex_grey8_stored.png": (9, 6), "tex_grey1.png": (13, 6), "tex_grey8.tga": (9, 6), "tex_grey.jpg": (21, 19),
               "tex_proggrey.jpg": (20, 21), "tex_max65535.pgm": (9, 6), "tex_grey2.png": (13, 6), "tex_grey4.png": (13, 6), "tex_grey16.png": (9, 6),
               "tex_grey4_adam7.png": (23, 19)}  # name -> (width, height)


def bitmap_fixtures():
    import struct
    import zlib
    import numpy as np
    rng = np.random.default_rng(23)
    fx = {}
    w, h = 9, 6
    rgb = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    grey = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    fx["tex_rgb8.png"] = _png(w, h, 8, 2, [rgb[y].tobytes() for y in range(h)])
    fx["tex_rgba8_fixed.png"] = _png(w, h, 8, 6, [rgba[y].tobytes() for y in range(h)], strategy=zlib.Z_FIXED)
    fx["tex_grey8_stored.png"] = _png(w, h, 8, 0, [grey[y].tobytes() for y in range(h)], level=0)
    ga = rng.integers(0, 256, size=(h, w, 2), dtype=np.uint8)
    fx["tex_greyalpha8.png"] = _png(w, h, 8, 4, [ga[y].tobytes() for y in range(h)])
    rgb16 = rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16)
    fx["tex_rgb16.png"] = _png(w, h, 16, 2, [rgb16[y].astype(">u2").tobytes() for y in range(h)], split_idat=3)
    big = rng.integers(0, 256, size=(19, 23, 3), dtype=np.uint8)
    fx["tex_rgb8_adam7.png"] = _png(23, 19, 8, 2, [big[y].tobytes() for y in range(19)], interlace=True)
    pal = rng.integers(0, 256, size=(16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, size=(h, w), dtype=np.uint8)
    rows4 = []
    for y in range(h):
        v = list(idx[y]) + [0] * (w % 2)
        rows4.append(bytes((v[i] << 4) | v[i + 1] for i in range(0, len(v), 2)))
    fx["tex_pal4.png"] = _png(w, h, 4, 3, rows4, palette=pal.reshape(-1).tolist())
    fx["tex_pal4_trns.png"] = _png(w, h, 4, 3, rows4, palette=pal.reshape(-1).tolist(), trns=[0, 128, 255, 7])
    bits = rng.integers(0, 2, size=(h, 13), dtype=np.uint8)
    rows1 = [bytes(np.packbits(bits[y]).tolist()) for y in range(h)]
    fx["tex_grey1.png"] = _png(13, h, 1, 0, rows1)
    noisy = rng.integers(0, 256, size=(40, 64, 3), dtype=np.uint8)
    noisy[:, :, 1] = (np.arange(64)[None, :] * 3 + np.arange(40)[:, None]) & 255   # compressible channel: long matches + literals
    fx["tex_rgb8_big.png"] = _png(64, 40, 8, 2, [noisy[y].tobytes() for y in range(40)])

    def bmp(w, h, bpp, rows_bgr, palette=None, top_down=False, v4=False):
        stride = ((w * bpp + 31) // 32) * 4
        body = b"".join(r + b"\0" * (stride - len(r)) for r in (rows_bgr if top_down else rows_bgr[::-1]))
        pal = b"" if palette is None else b"".join(bytes([c[2], c[1], c[0], 0]) for c in palette)
        hsz = 108 if v4 else 40
        off = 14 + hsz + len(pal)
        dib = struct.pack("<IiiHHIIiiII", hsz, w, -h if top_down else h, 1, bpp, 3 if v4 else 0, len(body), 2835, 2835, 0 if palette is None else len(palette), 0)
        if v4:
            dib += struct.pack("<IIII", 0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000) + b"BGRs" + b"\0" * 48
        return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + dib + pal + body
    fx["tex_24.bmp"] = bmp(w, h, 24, [rgb[y, :, ::-1].tobytes() for y in range(h)])
    fx["tex_24_topdown.bmp"] = bmp(w, h, 24, [rgb[y, :, ::-1].tobytes() for y in range(h)], top_down=True)
    bgra = rgba[:, :, [2, 1, 0, 3]]
    fx["tex_32_v4.bmp"] = bmp(w, h, 32, [bgra[y].tobytes() for y in range(h)], v4=True)
    pal8 = rng.integers(0, 256, size=(200, 3), dtype=np.uint8)
    idx8 = rng.integers(0, 200, size=(h, w), dtype=np.uint8)
    fx["tex_pal8.bmp"] = bmp(w, h, 8, [idx8[y].tobytes() for y in range(h)], palette=[tuple(int(x) for x in c) for c in pal8])

    def tga(w, h, bpp, rows, kind, top_left=False, rle=False):
        px = bpp // 8
        data = b"".join(rows if top_left else rows[::-1])
        if rle:
            out = bytearray()
            i, n = 0, w * h
            while i < n:
                run = 1
                while i + run < n and run < 128 and data[(i + run) * px:(i + run + 1) * px] == data[i * px:(i + 1) * px]:
                    run += 1
                if run > 1:
                    out.append(128 | (run - 1))
                    out += data[i * px:(i + 1) * px]
                    i += run
                else:
                    lit = 1
                    while i + lit < n and lit < 128 and data[(i + lit) * px:(i + lit + 1) * px] != data[(i + lit - 1) * px:(i + lit) * px]:
                        lit += 1
                    out.append(lit - 1)
                    out += data[i * px:(i + lit) * px]
                    i += lit
            data = bytes(out)
        ident = b"seeded"
        return struct.pack("<BBBHHBHHHHBB", len(ident), 0, kind + (8 if rle else 0), 0, 0, 0, 0, 0, w, h, bpp, (0x20 if top_left else 0) | (8 if bpp == 32 else 0)) + ident + data
    fx["tex_24.tga"] = tga(w, h, 24, [rgb[y, :, ::-1].tobytes() for y in range(h)], 2)
    flat = rgba.copy()
    flat[2:4, 1:7] = flat[2, 1]   # runs for the run-length coder
    fx["tex_32_rle_topleft.tga"] = tga(w, h, 32, [flat[y][:, [2, 1, 0, 3]].tobytes() for y in range(h)], 2, top_left=True, rle=True)
    fx["tex_grey8.tga"] = tga(w, h, 8, [grey[y].tobytes() for y in range(h)], 3)

    # JPEG: written by Pillow's libjpeg (any encoder would do: the known answers come from the reference's decoder).  Smooth
    # content + noise so that every frequency band carries something; 21 x 19 texels are covered completely by the 21 x 21 grid
    # of sample points, two MCUs across and down at every subsampling.
    import io
    try:
        from PIL import Image
    except ImportError as e:  # only needed to REGENERATE: the committed fixtures and their known answers stay valid without it
        raise SystemExit("oracle/make_golden.py: Pillow is needed to write the JPEG / GIF fixtures again (%s)" % e)

    def jpeg(name, size, mode, **kw):
        wj, hj = size
        yy, xx = np.mgrid[0:hj, 0:wj]
        ch = {"RGB": 3, "L": 1, "CMYK": 4}[mode]
        a = np.stack([(xx * 7 + yy * 3 + 40 * np.sin(xx / 3.0 + k)) % 256 for k in range(ch)], -1) + rng.integers(-20, 20, (hj, wj, ch))
        a = np.clip(a, 0, 255).astype(np.uint8)
        buf = io.BytesIO()
        Image.fromarray(a[:, :, 0] if ch == 1 else a, mode=mode).save(buf, format="JPEG", **kw)
        fx[name] = buf.getvalue()
    jpeg("tex_444.jpg", (21, 19), "RGB", quality=90, subsampling=0)
    jpeg("tex_422.jpg", (21, 19), "RGB", quality=85, subsampling=1)
    jpeg("tex_420.jpg", (21, 19), "RGB", quality=75, subsampling=2)
    jpeg("tex_411.jpg", (21, 19), "RGB", quality=80, subsampling="4:1:1")
    jpeg("tex_grey.jpg", (21, 19), "L", quality=80)
    jpeg("tex_prog420.jpg", (21, 19), "RGB", quality=70, subsampling=2, progressive=True)
    jpeg("tex_prog444_opt.jpg", (21, 19), "RGB", quality=95, subsampling=0, progressive=True, optimize=True)
    jpeg("tex_proggrey.jpg", (20, 21), "L", quality=50, progressive=True)
    jpeg("tex_q10.jpg", (21, 19), "RGB", quality=10, subsampling=2)
    jpeg("tex_rst.jpg", (64, 40), "RGB", quality=60, subsampling=2, restart_marker_blocks=3)
    jpeg("tex_cmyk.jpg", (21, 19), "CMYK", quality=90)
    jpeg("tex_1x1.jpg", (1, 1), "RGB", quality=90, subsampling=2)
    jpeg("tex_2x1.jpg", (2, 1), "RGB", quality=90, subsampling=2)
    jpeg("tex_1x9.jpg", (1, 9), "RGB", quality=90, subsampling=1)

    # GIF: Pillow's encoder for whole-canvas images (palette, interlaced, transparent index), and a hand-assembled file whose first
    # image covers only part of the canvas (background index, local colour table, graphic control extension)
    def pil_gif(name, size, **kw):
        wg, hg = size
        im = Image.fromarray(rng.integers(0, 64, (hg, wg), dtype=np.uint8), mode="P")
        im.putpalette(rng.integers(0, 256, 64 * 3, dtype=np.uint8).tolist())
        buf = io.BytesIO()
        im.save(buf, format="GIF", **kw)
        fx[name] = buf.getvalue()
    pil_gif("tex_pal.gif", (9, 6))
    pil_gif("tex_interlaced.gif", (21, 19), interlace=True)
    pil_gif("tex_transparent.gif", (9, 6), transparency=5)

    def raw_gif(wc, hc, bg, table, x0, y0, wi, hi, indices, local=None, transparent=None, interlace=False):
        bits = max(2, (len(local or table) - 1).bit_length())
        out = bytearray(b"GIF89a" + struct.pack("<HHBBB", wc, hc, 0x80 | ((len(table).bit_length() - 2) & 7), bg, 0))
        out += bytes(c for e in table for c in e)
        if transparent is not None:
            out += bytes([0x21, 0xF9, 4, 1, 0, 0, transparent, 0])
        out += b"\x21\xFE\x06seeded\x00"  # a comment extension to skip
        out += b"\x2C" + struct.pack("<HHHHB", x0, y0, wi, hi, (0x40 if interlace else 0) | ((0x80 | ((len(local).bit_length() - 2) & 7)) if local else 0))
        if local:
            out += bytes(c for e in local for c in e)
        clear, codes, size = 1 << bits, [], bits + 1
        rows = list(range(hi))
        if interlace:
            rows = list(range(0, hi, 8)) + list(range(4, hi, 8)) + list(range(2, hi, 4)) + list(range(1, hi, 2))
        stream = [indices[r][c] for r in rows for c in range(wi)]
        # literal codes only, a clear code often enough that the code size never grows (valid LZW, no compression)
        per = (1 << size) - clear - 3
        for i, v in enumerate(stream):
            if i % per == 0:
                codes.append(clear)
            codes.append(int(v))
        codes.append(clear + 1)
        acc = n = 0
        data = bytearray()
        for c in codes:
            acc |= c << n
            n += size
            while n >= 8:
                data.append(acc & 255)
                acc >>= 8
                n -= 8
        if n:
            data.append(acc & 255)
        out.append(bits)
        for i in range(0, len(data), 200):
            out.append(len(data[i:i + 200]))
            out += data[i:i + 200]
        return bytes(out + b"\x00\x3B")
    tab8 = [tuple(int(x) for x in rng.integers(0, 256, 3)) for _ in range(8)]
    loc16 = [tuple(int(x) for x in rng.integers(0, 256, 3)) for _ in range(16)]
    fx["tex_partial_bg.gif"] = raw_gif(9, 6, 3, tab8, 2, 1, 5, 3, rng.integers(0, 8, (3, 5)))
    fx["tex_partial_local_transparent.gif"] = raw_gif(9, 6, 2, tab8, 1, 1, 7, 4, rng.integers(0, 16, (4, 7)), local=loc16, transparent=6)
    fx["tex_partial_interlaced.gif"] = raw_gif(21, 19, 0, tab8, 3, 2, 15, 13, rng.integers(0, 8, (13, 15)), interlace=True)

    # Radiance HDR: flat texels (narrow image), run-length coded scanlines (runs and literals in each of the four planes), and a
    # wide image stored flat; exponents around 128 so that the tone curve's whole range is used
    def hdr(wh, hh, coded, magic=b"#?RADIANCE"):
        q = np.concatenate([rng.integers(0, 256, (hh, wh, 3)), rng.integers(120, 131, (hh, wh, 1))], -1).astype(np.uint8)
        q[0, 0, 3] = 0  # an exponent of zero is black whatever the mantissas say
        q[hh // 2, :, 3] = 127  # a row that compresses
        q[hh // 2, : wh // 2, :3] = q[hh // 2, 0, :3]
        out = bytearray(magic + b"\nEXPOSURE=1.0\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (hh, wh))
        for y in range(hh):
            if not coded:
                out += q[y].tobytes()
                continue
            out += bytes([2, 2, wh >> 8, wh & 255])
            for k in range(4):
                row, i = q[y, :, k], 0
                while i < wh:
                    run = 1
                    while i + run < wh and run < 127 and row[i + run] == row[i]:
                        run += 1
                    if run >= 3:
                        out += bytes([128 + run, int(row[i])])
                        i += run
                    else:
                        lit = min(wh - i, 5)
                        out += bytes([lit]) + row[i:i + lit].tobytes()
                        i += lit
        return bytes(out)
    fx["tex_flat_narrow.hdr"] = hdr(7, 6, False)
    fx["tex_rle.hdr"] = hdr(21, 19, True)
    fx["tex_flat_wide.hdr"] = hdr(12, 5, False, magic=b"#?RGBE")

    # Photoshop PSD (merged image): 8-bit raw RGB, 16-bit raw RGB, PackBits RGBA whose colours are matted on white
    def psd(wp, hp, planes, depth=8, packbits=False):
        out = bytearray(b"8BPS" + struct.pack(">H6xHIIHH", 1, len(planes), hp, wp, depth, 3) + struct.pack(">III", 0, 0, 0))
        out += struct.pack(">H", 1 if packbits else 0)
        if not packbits:
            for pl in planes:
                out += pl.astype(">u2" if depth == 16 else np.uint8).tobytes()
            return bytes(out)
        coded = []
        for pl in planes:
            for row in pl:
                c, i = bytearray(), 0
                while i < wp:
                    run = 1
                    while i + run < wp and run < 128 and row[i + run] == row[i]:
                        run += 1
                    if run >= 2:
                        c += bytes([257 - run, int(row[i])])
                        i += run
                    else:
                        lit = min(wp - i, 4)
                        c += bytes([lit - 1]) + bytes(int(v) for v in row[i:i + lit])
                        i += lit
                coded.append(bytes(c))
        out += b"".join(struct.pack(">H", len(c)) for c in coded) + b"".join(coded)
        return bytes(out)
    fx["tex_rgb8.psd"] = psd(9, 6, [rng.integers(0, 256, (6, 9)) for _ in range(3)])
    fx["tex_rgb16.psd"] = psd(9, 6, [rng.integers(0, 65536, (6, 9)) for _ in range(3)], depth=16)
    alpha = rng.integers(0, 256, (6, 9))
    alpha[0, :3] = (0, 255, 128)
    alpha[2, 2:7] = 200
    matted = []
    for _ in range(3):
        colour = rng.integers(0, 256, (6, 9))
        colour[2, 2:7] = colour[2, 2]
        matted.append((colour * alpha + 255 * (255 - alpha) + 127) // 255)  # over white: the un-matting stays inside 0..255
    fx["tex_rgba8_packbits.psd"] = psd(9, 6, matted + [alpha], packbits=True)

    # Softimage PIC: raw RGB; mixed run-length RGB chained to a pure run-length alpha packet; long 16-bit runs
    def pic(wq, hq, packets, rows):
        out = bytearray(b"\x53\x80\xF6\x34" + struct.pack(">f", 3.71) + b"seeded".ljust(80, b"\0") + b"PICT" + struct.pack(">HHfHH", wq, hq, 1.0, 3, 0))
        for i, (kind, mask) in enumerate(packets):
            out += bytes([1 if i + 1 < len(packets) else 0, 8, kind, mask])
        for y in range(hq):
            for (kind, mask), line in zip(packets, rows):
                px = [bytes(int(v) for v in t) for t in line[y]]  # per texel: the bytes of the packet's channels, in R G B A order
                if kind == 0:
                    out += b"".join(px)
                    continue
                i = 0
                while i < wq:
                    run = 1
                    while i + run < wq and px[i + run] == px[i] and run < (255 if kind == 1 else 400):
                        run += 1
                    if kind == 1:
                        out += bytes([run]) + px[i]
                        i += run
                    elif run >= 130:
                        out += bytes([128]) + struct.pack(">H", run) + px[i]
                        i += run
                    elif run >= 2:
                        out += bytes([127 + run]) + px[i]
                        i += run
                    else:
                        lit = min(wq - i, 3)
                        out += bytes([lit - 1]) + b"".join(px[i:i + lit])
                        i += lit
        return bytes(out)
    rgbp = rng.integers(0, 256, (6, 9, 3))
    fx["tex_raw.pic"] = pic(9, 6, [(0, 0xE0)], [rgbp])
    rgbq = rng.integers(0, 256, (6, 9, 3))
    rgbq[1:4, 2:8] = rgbq[1, 2]
    alq = rng.integers(0, 256, (6, 9, 1))
    alq[:, 3:7] = 77
    fx["tex_mixed_alpha.pic"] = pic(9, 6, [(2, 0xE0), (1, 0x10)], [rgbq, alq])
    wide = np.zeros((3, 300, 3), dtype=np.int64)
    wide[:, :, 0] = (np.arange(300) // 150) * 200
    wide[:, :, 1] = rng.integers(0, 256, (3, 1))
    wide[1, 290:, 2] = rng.integers(0, 256, 10)
    fx["tex_long_runs.pic"] = pic(300, 3, [(2, 0xE0)], [wide])

    # BMP with 16 bits per pixel and with channel masks (BI_BITFIELDS): 5-5-5 by default, 5-6-5 and 8-bit fields at odd places behind
    # a 40-byte header, 4-4-4-4 with alpha in a V4 header; 4-bit and 1-bit palettes
    def bmp_masks(wb, hb, bpp, words, masks=None, v4_alpha=None):
        stride = ((wb * bpp + 31) // 32) * 4
        body = b"".join(struct.pack("<%d%s" % (wb, "H" if bpp == 16 else "I"), *[int(v) for v in row]).ljust(stride, b"\0") for row in words[::-1])
        if v4_alpha is not None:
            dib = struct.pack("<IiiHHIIiiII", 108, wb, hb, 1, bpp, 3, len(body), 2835, 2835, 0, 0) + struct.pack("<IIII", *masks, v4_alpha) + b"BGRs" + b"\0" * 48
            extra = b""
        else:
            dib = struct.pack("<IiiHHIIiiII", 40, wb, hb, 1, bpp, 3 if masks else 0, len(body), 2835, 2835, 0, 0)
            extra = struct.pack("<III", *masks) if masks else b""
        off = 14 + len(dib) + len(extra)
        return b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + dib + extra + body
    fx["tex_555.bmp"] = bmp_masks(w, h, 16, rng.integers(0, 1 << 15, (h, w)))
    fx["tex_565.bmp"] = bmp_masks(w, h, 16, rng.integers(0, 1 << 16, (h, w)), masks=(0xF800, 0x07E0, 0x001F))
    fx["tex_odd_masks32.bmp"] = bmp_masks(w, h, 32, rng.integers(0, 1 << 32, (h, w), dtype=np.uint64), masks=(0x3FC00000, 0x000FE000, 0x000000E0))
    fx["tex_4444_v4.bmp"] = bmp_masks(w, h, 16, rng.integers(0, 1 << 16, (h, w)), masks=(0x0F00, 0x00F0, 0x000F), v4_alpha=0xF000)
    pal16 = [tuple(int(x) for x in c) for c in rng.integers(0, 256, (16, 3))]
    idx4 = rng.integers(0, 16, (h, w))
    fx["tex_pal4.bmp"] = bmp(w, h, 4, [bytes((int(r[i]) << 4) | (int(r[i + 1]) if i + 1 < w else 0) for i in range(0, w, 2)) for r in idx4], palette=pal16)
    bits1 = rng.integers(0, 2, (h, 13))
    fx["tex_pal1.bmp"] = bmp(13, h, 1, [bytes(np.packbits(r).tolist()) for r in bits1], palette=pal16[:2])

    # TGA beyond true colour: colour-mapped (8-bit indices into 24-bit entries, one index outside the palette; run-length coded
    # 16-bit indices into 5-5-5 entries behind a "first entry" field), 16-bit 5-5-5 true colour, grey + alpha, and the right-to-left
    # descriptor bit (which the reference's decoder ignores)
    def tga2(wt, ht, kind, bpp, texels, descriptor=0, cmap=None, cmap_bits=0, cmap_first=0, rle=False):
        px = bpp // 8
        data = b"".join(texels)
        if rle:
            out, i, n = bytearray(), 0, wt * ht
            while i < n:
                run = 1
                while i + run < n and run < 128 and texels[i + run] == texels[i]:
                    run += 1
                if run > 1:
                    out.append(128 | (run - 1)); out += texels[i]; i += run
                else:
                    lit = min(3, n - i)
                    out.append(lit - 1); out += b"".join(texels[i:i + lit]); i += lit
            data = bytes(out)
        head = struct.pack("<BBBHHBHHHHBB", 0, 1 if cmap else 0, kind + (8 if rle else 0), cmap_first, len(cmap) if cmap else 0, cmap_bits, 0, 0, wt, ht, bpp, descriptor)
        return head + (b"\0" * cmap_first + b"".join(cmap) if cmap else b"") + data
    cm24 = [bytes(int(x) for x in c) for c in rng.integers(0, 256, (40, 3))]
    idx = [bytes([int(v)]) for v in rng.integers(0, 40, w * h)]
    idx[7] = bytes([200])  # outside the palette: entry 0
    fx["tex_mapped8.tga"] = tga2(w, h, 1, 8, idx, cmap=cm24, cmap_bits=24)
    cm16 = [struct.pack("<H", int(v)) for v in rng.integers(0, 1 << 16, 300)]
    idx16 = [struct.pack("<H", int(v)) for v in rng.integers(0, 300, w * h)]
    idx16[10:16] = [idx16[10]] * 6
    fx["tex_mapped16_rle.tga"] = tga2(w, h, 1, 16, idx16, descriptor=0x20, cmap=cm16, cmap_bits=16, cmap_first=4, rle=True)
    fx["tex_555.tga"] = tga2(w, h, 2, 16, [struct.pack("<H", int(v)) for v in rng.integers(0, 1 << 16, w * h)])
    fx["tex_greyalpha16.tga"] = tga2(w, h, 3, 16, [bytes(int(x) for x in c) for c in rng.integers(0, 256, (w * h, 2))], descriptor=8)
    fx["tex_24_right_to_left.tga"] = tga2(w, h, 2, 24, [bytes(int(x) for x in c) for c in rng.integers(0, 256, (w * h, 3))], descriptor=0x10)

    # PNM with a maximum other than 255: below, the bytes are taken as they are; above, samples are two bytes
    deep = rng.integers(0, 1001, (h, w, 3))
    fx["tex_max1000.ppm"] = b"P6\n# seeded\n%d %d\n1000\n" % (w, h) + deep.astype(">u2").tobytes()
    fx["tex_max65535.pgm"] = b"P5 %d %d 65535\n" % (w, h) + rng.integers(0, 65536, (h, w)).astype(">u2").tobytes()
    fx["tex_max100.ppm"] = b"P6 %d %d 100 " % (w, h) + rng.integers(0, 101, (h, w, 3)).astype(np.uint8).tobytes()

    # PNG, the remaining depth x colour-type combinations, colour-key transparency, and Adam7 below 8 bits
    def packed(vals, depth):
        out = []
        for row in vals:
            acc = bytearray((len(row) * depth + 7) // 8)
            for i, v in enumerate(row):
                acc[(i * depth) >> 3] |= int(v) << (8 - depth - ((i * depth) & 7))
            out.append(bytes(acc))
        return out
    fx["tex_grey2.png"] = _png(13, h, 2, 0, packed(rng.integers(0, 4, (h, 13)), 2))
    fx["tex_grey4.png"] = _png(13, h, 4, 0, packed(rng.integers(0, 16, (h, 13)), 4))
    g16 = rng.integers(0, 65536, (h, w))
    fx["tex_grey16.png"] = _png(w, h, 16, 0, [g16[y].astype(">u2").tobytes() for y in range(h)])
    fx["tex_greyalpha16.png"] = _png(w, h, 16, 4, [rng.integers(0, 65536, (w, 2))[:, :].astype(">u2").tobytes() for _ in range(h)])
    fx["tex_rgba16.png"] = _png(w, h, 16, 6, [rng.integers(0, 65536, (w, 4)).astype(">u2").tobytes() for _ in range(h)])
    pal256 = rng.integers(0, 256, 256 * 3).tolist()
    fx["tex_pal8.png"] = _png(w, h, 8, 3, [rng.integers(0, 256, w, dtype=np.uint8).tobytes() for _ in range(h)], palette=pal256)
    fx["tex_pal1.png"] = _png(13, h, 1, 3, packed(rng.integers(0, 2, (h, 13)), 1), palette=pal256[:6])
    fx["tex_pal2_adam7.png"] = _png(23, 19, 2, 3, packed(rng.integers(0, 4, (19, 23)), 2), palette=pal256[:12], interlace=True, trns=[255, 0, 90])
    fx["tex_grey4_adam7.png"] = _png(23, 19, 4, 0, packed(rng.integers(0, 16, (19, 23)), 4), interlace=True)
    key = rng.integers(0, 4, (h, w, 3)) * 85  # few distinct colours: the colour key below names one that occurs
    fx["tex_rgb8_key.png"] = _png(w, h, 8, 2, [key[y].astype(np.uint8).tobytes() for y in range(h)], trns=[0, 85, 0, 170, 0, 0])
    gk = rng.integers(0, 4, (h, w)) * 85
    fx["tex_grey8_key.png"] = _png(w, h, 8, 0, [gk[y].astype(np.uint8).tobytes() for y in range(h)], trns=[0, 170])
    g16k = rng.integers(0, 3, (h, w)) * 30000
    fx["tex_grey16_key.png"] = _png(w, h, 16, 0, [g16k[y].astype(">u2").tobytes() for y in range(h)], trns=[0x75, 0x30])
    return fx


if __name__ == "__main__":
    sys.exit(main())
