"""FITS binary-table light curves -> ragged device arrays ("FITS -> ragged device arrays").

The reference reads a light-curve file with astropy (``Table.read(hdulist[ext])``, reference
src/lightkurve/io/generic.py:21-207), builds Time / Quantity / Table objects per file, then drops NaN times (:98-101) and
the cadences whose quality flag hits the bitmask (io/kepler.py:49-53, io/tess.py:45-48, utils.py:79-115).  Here the host
only parses the two headers (no astropy in the product interpreter) and hands the table's raw big-endian bytes to
``lk_fits_unpack_batch``: byte swapping, column extraction, the NaN-time and quality filters and the packing of the whole
batch into (time, flux, flux_err, quality, n_off) happen on the GPU.

    batch = LightCurveBatch.from_fits(paths)                      # ingest.py; Kepler / TESS defaults like lk.read()
    tab = read_fits_table(path); tab.columns["time"]              # (byte offset, TFORM letter, repeat)
"""
import gzip

import numpy as np

__all__ = ["read", "read_fits_table", "read_fits_image", "FitsTable", "lightcurve_columns", "pixel_columns", "QUALITY_OPTIONS"]

# reference src/lightkurve/utils.py: KeplerQualityFlags.OPTIONS (:190-195), TessQualityFlags.OPTIONS (:270-275)
QUALITY_OPTIONS = {
    "kepler": {"none": 0, "default": 1130799, "hard": 1664431, "hardest": 2096639},
    "tess": {"none": 0, "default": 17087, "hard": 24319, "hardest": 65535},
}
_TFORM_BYTES = {"L": 1, "B": 1, "I": 2, "J": 4, "K": 8, "A": 1, "E": 4, "D": 8, "C": 8, "M": 16, "P": 8, "Q": 16}
BLOCK = 2880


def _parse_value(text):
    """Value field of a header card (FITS standard 4.0 §4.2): string, logical, integer or real."""
    text = text.strip()
    if not text:
        return None
    if text[0] == "'":
        out, i = [], 1
        while i < len(text):
            if text[i] == "'":
                if i + 1 < len(text) and text[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(text[i])
            i += 1
        return "".join(out).rstrip()
    text = text.split("/", 1)[0].strip()
    if text in ("T", "F"):
        return text == "T"
    try:
        return int(text)
    except ValueError:
        pass
    try:
        return float(text.replace("D", "E").replace("d", "e"))
    except ValueError:
        return text


def _read_header(buf, pos):
    """Cards from ``pos`` up to END -> (dict, position of the first data byte).  Raises on a truncated header."""
    hdr = {}
    while True:
        if pos + BLOCK > len(buf):
            raise OSError("truncated FITS header (no END card)")
        block = bytes(buf[pos:pos + BLOCK]).decode("ascii", "replace")
        pos += BLOCK
        for c in range(0, BLOCK, 80):
            card = block[c:c + 80]
            key = card[:8].strip()
            if key == "END":
                return hdr, pos
            if card[8:10] == "= " and key and key not in hdr:
                hdr[key] = _parse_value(card[10:])


class FitsTable(object):
    """One BINTABLE extension: ``raw`` (uint8, n_rows x row_bytes, the file's bytes untouched), ``columns`` (lower-case
    name -> (byte offset in the row, TFORM letter, repeat)), ``header`` (extension) and ``primary`` (HDU 0)."""

    def __init__(self, raw, columns, header, primary, path=None, dims=None):
        self.raw, self.columns, self.header, self.primary, self.path = raw, columns, header, primary, path
        self.dims = dims or {}   # lower-case name -> numpy shape of one cell (from TDIMn), vector columns only

    @property
    def n_rows(self):
        return self.raw.shape[0]

    @property
    def row_bytes(self):
        return self.raw.shape[1]


def read_fits_table(path, ext=1):
    """Parse HDU 0's header and binary-table extension ``ext`` (1-based like astropy's hdulist[ext]) of a FITS file
    (optionally gzip-compressed).  No value is converted: ``raw`` is a view of the table's bytes."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as fh:
        buf = np.frombuffer(fh.read(), dtype=np.uint8)
    if buf.size < BLOCK or bytes(buf[:6]) != b"SIMPLE":
        raise OSError("%s is not a FITS file" % path)
    primary, pos = _read_header(buf, 0)
    nax = int(primary.get("NAXIS", 0))
    size = 0
    if nax > 0:
        size = abs(int(primary.get("BITPIX", 8))) // 8
        for i in range(1, nax + 1):
            size *= int(primary["NAXIS%d" % i])
    pos += (size + BLOCK - 1) // BLOCK * BLOCK
    hdr = None
    for _ in range(int(ext)):
        hdr, pos = _read_header(buf, pos)
        size = abs(int(hdr.get("BITPIX", 8))) // 8
        for i in range(1, int(hdr.get("NAXIS", 0)) + 1):
            size *= int(hdr["NAXIS%d" % i])
        size = (size if int(hdr.get("NAXIS", 0)) > 0 else 0) + int(hdr.get("PCOUNT", 0))
        data_pos = pos
        pos += (size + BLOCK - 1) // BLOCK * BLOCK
    if hdr is None or str(hdr.get("XTENSION", "")).strip() != "BINTABLE":
        raise OSError("extension %s of %s is not a binary table" % (ext, path))
    row_bytes, n_rows = int(hdr["NAXIS1"]), int(hdr["NAXIS2"])
    if data_pos + row_bytes * n_rows > buf.size:
        raise OSError("%s: the table is truncated (%d of %d bytes)" % (path, buf.size - data_pos, row_bytes * n_rows))
    columns, dims, off = {}, {}, 0
    for i in range(1, int(hdr["TFIELDS"]) + 1):
        form = str(hdr["TFORM%d" % i]).strip()
        j = 0
        while j < len(form) and form[j].isdigit():
            j += 1
        repeat = int(form[:j]) if j else 1
        letter = form[j].upper()
        nbytes = (repeat + 7) // 8 if letter == "X" else repeat * _TFORM_BYTES[letter]
        name = str(hdr.get("TTYPE%d" % i, "col%d" % i)).strip().lower()
        scaled = float(hdr.get("TSCAL%d" % i, 1.0)) != 1.0 or float(hdr.get("TZERO%d" % i, 0.0)) != 0.0
        columns.setdefault(name, (off, letter, repeat, scaled))
        tdim = hdr.get("TDIM%d" % i)
        if tdim and name not in dims:  # '(ncol,nrow)' in Fortran order -> numpy shape (nrow, ncol)
            dims[name] = tuple(int(x) for x in str(tdim).strip().strip("()").split(","))[::-1]
        off += nbytes
    if off != row_bytes:
        raise OSError("%s: TFORM widths add up to %d bytes, NAXIS1 says %d" % (path, off, row_bytes))
    raw = buf[data_pos:data_pos + row_bytes * n_rows].reshape(n_rows, row_bytes)
    return FitsTable(raw, columns, hdr, primary, path=str(path), dims=dims)


def read_fits_image(path, ext):
    """Image extension ``ext`` (1-based) as a native-endian ndarray, or None if the file has no such HDU / it is not an
    image.  Used for the aperture extension of target-pixel files (a few hundred integers: decoded on the host)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as fh:
        buf = np.frombuffer(fh.read(), dtype=np.uint8)
    pos, hdr = 0, None
    try:
        for i in range(int(ext) + 1):
            hdr, pos = _read_header(buf, pos)
            nax = int(hdr.get("NAXIS", 0))
            size = abs(int(hdr.get("BITPIX", 8))) // 8 if nax > 0 else 0
            for k in range(1, nax + 1):
                size *= int(hdr["NAXIS%d" % k])
            size += int(hdr.get("PCOUNT", 0))
            data_pos = pos
            pos += (size + BLOCK - 1) // BLOCK * BLOCK
    except OSError:
        return None
    if str(hdr.get("XTENSION", "IMAGE")).strip() != "IMAGE" or int(hdr.get("NAXIS", 0)) == 0:
        return None
    dt = {8: "u1", 16: ">i2", 32: ">i4", 64: ">i8", -32: ">f4", -64: ">f8"}[int(hdr["BITPIX"])]
    shape = tuple(int(hdr["NAXIS%d" % k]) for k in range(int(hdr["NAXIS"]), 0, -1))
    n = int(np.prod(shape))
    arr = np.frombuffer(bytes(buf[data_pos:data_pos + n * np.dtype(dt).itemsize]), dtype=dt).reshape(shape)
    return arr.astype(arr.dtype.newbyteorder("=")) * hdr.get("BSCALE", 1) + hdr.get("BZERO", 0) \
        if ("BSCALE" in hdr or "BZERO" in hdr) else arr.astype(arr.dtype.newbyteorder("="))


_CODES = {"D": 0, "E": 1, "J": 2, "K": 3, "I": 4, "B": 5}


def lightcurve_columns(tab, flux_column=None, quality_bitmask="default", mission=None):
    """Which bytes of a row are (time, flux, flux_err, quality) and which bitmask applies — the reference's choices:
    Kepler/K2 files: flux_column 'pdcsap_flux', quality column 'sap_quality' (io/kepler.py:10-44); TESS: 'pdcsap_flux',
    'quality' (io/tess.py:10-40); anything else: 'flux', 'quality' (io/generic.py:21-32).  flux_err falls back to
    '<flux_column>_err' (generic.py:160-165).  Returns (desc[10] int32, bitmask int, mission)."""
    if mission is None:
        tel = str(tab.primary.get("TELESCOP", tab.header.get("TELESCOP", ""))).strip().lower()
        mission = "kepler" if tel in ("kepler", "k2") else ("tess" if tel == "tess" else "generic")
    cols = tab.columns
    time_col = "time" if "time" in cols else ("t" if "t" in cols else None)   # generic.py:89-91
    if time_col is None:
        raise KeyError("%s has no TIME column" % tab.path)
    if flux_column is None:
        flux_column = "pdcsap_flux" if mission in ("kepler", "tess") else "flux"
    flux_column = flux_column.lower()
    # generic.py:157-165: an existing FLUX / FLUX_ERR column is kept as it is; otherwise flux = <flux_column> and
    # flux_err = <flux_column>_err
    use_flux = "flux" if "flux" in cols else flux_column
    if use_flux not in cols:
        raise KeyError("%s has no %s column" % (tab.path, flux_column.upper()))
    err_col = "flux_err" if "flux_err" in cols else flux_column + "_err"
    flux_column = use_flux
    qual_col = "sap_quality" if mission == "kepler" else "quality"
    if mission == "generic":
        # the generic reader applies no quality mask at all (io/generic.py); an explicit integer still works here
        bitmask = int(quality_bitmask) if isinstance(quality_bitmask, (int, np.integer)) else 0
    elif isinstance(quality_bitmask, str):
        opts = QUALITY_OPTIONS[mission]
        if quality_bitmask not in opts:
            raise ValueError("quality_bitmask='{}' is not supported, expected one of {}".format(quality_bitmask, tuple(opts)))
        bitmask = opts[quality_bitmask]
    else:
        bitmask = 0 if quality_bitmask is None else int(quality_bitmask)

    def field(name, allowed):
        if name not in cols:
            return -1, 0
        off, letter, repeat, scaled = cols[name]
        if letter not in allowed or repeat != 1:
            raise NotImplementedError("column %s of %s has TFORM %d%s" % (name.upper(), tab.path, repeat, letter))
        if scaled:
            raise NotImplementedError("column %s of %s is scaled (TSCAL/TZERO)" % (name.upper(), tab.path))
        return off, _CODES[letter]

    ot, ct = field(time_col, "DE")
    of, cf = field(flux_column, "DE")
    oe, ce = field(err_col, "DE")
    oq, cq = field(qual_col, "JKIB")
    desc = np.array([tab.row_bytes, tab.n_rows, ot, ct, of, cf, oe, ce, oq, cq], dtype=np.int32)
    return desc, int(bitmask), mission


def pixel_columns(tab, columns=("flux", "flux_err", "flux_bkg"), quality_bitmask="default", mission=None):
    """Where a target-pixel file keeps TIME, QUALITY and its pixel cubes, and which cadences the reference keeps:
    quality_mask = (QUALITY & bitmask) == 0 (targetpixelfile.py:2120-2122 Kepler / K2, :2794-2797 TESS), TESS also drops
    NaN times unless the bitmask is 0 / 'none' (:2798-2801).  Returns a dict for ``_capi.fits_unpack_cube``."""
    if mission is None:
        tel = str(tab.primary.get("TELESCOP", tab.header.get("TELESCOP", ""))).strip().lower()
        mission = "tess" if tel == "tess" else "kepler"
    opts = QUALITY_OPTIONS[mission]
    if isinstance(quality_bitmask, str):
        if quality_bitmask not in opts:
            raise ValueError("quality_bitmask='{}' is not supported, expected one of {}".format(quality_bitmask, tuple(opts)))
        bitmask = opts[quality_bitmask]
    else:
        bitmask = 0 if quality_bitmask is None else int(quality_bitmask)
    cols = tab.columns
    for need in ("time", "quality", columns[0]):
        if need not in cols:
            raise KeyError("%s has no %s column" % (tab.path, need.upper()))
    ot, lt, rt, st = cols["time"]
    oq, lq, rq, sq = cols["quality"]
    if lt not in "DE" or lq not in "JKIB" or rt != 1 or rq != 1 or st or sq:
        raise NotImplementedError("TIME / QUALITY of %s have unsupported TFORMs" % tab.path)
    use, offs, npix, shape = [], [], None, None
    for name in columns:
        if name not in cols:
            continue
        off, letter, repeat, scaled = cols[name]
        if letter != "E" or scaled:
            raise NotImplementedError("pixel column %s of %s is not plain float32" % (name.upper(), tab.path))
        if npix is None:
            npix, shape = repeat, tab.dims.get(name, (repeat,))
        elif repeat != npix:
            raise ValueError("pixel columns of %s have different sizes" % tab.path)
        use.append(name), offs.append(off)
    keep_nan_time = mission != "tess" or bitmask == 0 or quality_bitmask == "none"
    return dict(off_time=ot, code_time=_CODES[lt], off_quality=oq, code_quality=_CODES[lq], bitmask=bitmask,
                keep_nan_time=keep_nan_time, columns=use, col_offsets=offs, npix=npix, shape=shape, mission=mission)


def read(path, quality_bitmask="default", flux_column=None, device=0):
    """``lightkurve.read(path)`` for local Kepler / K2 / TESS FITS products (reference src/lightkurve/io/read.py:32-145):
    a target-pixel file (its table's FLUX column is a pixel vector) comes back as a ``PixelCube``, a light-curve file as
    a ``LightCurve`` — with the reference's quality masking and NaN-time handling, decoded on the GPU.  Remote paths, the
    community readers (QLP, EVEREST, ...) and FoldedLightCurve files stay with the reference."""
    tab = read_fits_table(path, ext=1)
    flux = tab.columns.get("flux")
    if flux is not None and flux[2] > 1:
        from .correctors.pldcorrector import PixelCube
        return PixelCube.from_fits(path, quality_bitmask=quality_bitmask, device=device)
    from .ingest import LightCurveBatch
    batch = LightCurveBatch.from_fits([path], flux_column=flux_column, quality_bitmask=quality_bitmask, device=device)
    lc = batch[0]
    lc.quality = batch.quality
    return lc
