"""Re-code the entropy segment of a baseline JPEG with Huffman tables of the test's choosing.

Test data maker (pure Python, small images only): takes a single-scan baseline file without
restart markers and the coefficients it decodes to, and writes the same image with other
DHT segments.  Used to reach decoder paths that ordinary encoders never produce: AC tables
whose frequent symbols all have 11-bit codes spread over more 10-bit prefixes than the
kernels keep second-level tables for (canonical walk, decode.c:674-681)."""
import struct

import numpy as np

ZRL, EOB = 0xF0, 0x00


def _segments(data):
    """[(marker, start, end)] of the segments before the entropy data, and the SOS end."""
    assert data[:2] == b"\xff\xd8"
    pos, segs = 2, []
    while True:
        assert data[pos] == 0xFF, pos
        m = data[pos + 1]
        ln = struct.unpack(">H", data[pos + 2:pos + 4])[0]
        segs.append((m, pos, pos + 2 + ln))
        pos += 2 + ln
        if m == 0xDA:
            return segs, pos


def _canonical(symbols_by_len):
    """{symbol: (code, length)} for symbols_by_len[l] = symbols of code length l, in DHT order."""
    code, out = 0, {}
    for ln in range(1, 17):
        for s in symbols_by_len.get(ln, []):
            out[s] = (code, ln)
            code += 1
        code <<= 1
    return out


def _dht(tc, th, symbols_by_len):
    counts = bytes(len(symbols_by_len.get(ln, [])) for ln in range(1, 17))
    syms = bytes(s for ln in range(1, 17) for s in symbols_by_len.get(ln, []))
    body = bytes([(tc << 4) | th]) + counts + syms
    return b"\xff\xc4" + struct.pack(">H", 2 + len(body)) + body


def _block_symbols(blk, pred):
    """(dc category, dc bits, [(ac symbol, size, bits)]) of one block in zigzag order."""
    diff = int(blk[0]) - pred
    cat = abs(diff).bit_length()
    dbits = diff if diff >= 0 else diff - 1 + (1 << cat)
    ac, run = [], 0
    last = max((k for k in range(1, 64) if blk[k]), default=0)
    for k in range(1, last + 1):
        v = int(blk[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            ac.append((ZRL, 0, 0))
            run -= 16
        size = abs(v).bit_length()
        ac.append(((run << 4) | size, size, v if v >= 0 else v - 1 + (1 << size)))
        run = 0
    if last < 63:
        ac.append((EOB, 0, 0))
    return cat, dbits, ac


def long_code_ac_table(freq):
    """Code lengths for the 162 baseline AC symbols: the ten rarest get 4 bits, all others 11
    bits -- 152 codes over 76 ten-bit prefixes.  Returns symbols_by_len."""
    syms = [EOB, ZRL] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    order = sorted(syms, key=lambda s: (freq.get(s, 0), s))
    return {4: order[:10], 11: order[10:]}


def sixteen_bit_ac_table(freq):
    """The six rarest symbols get 3 bits, all others 16 bits."""
    syms = [EOB, ZRL] + [(r << 4) | s for r in range(16) for s in range(1, 11)]
    order = sorted(syms, key=lambda s: (freq.get(s, 0), s))
    return {3: order[:6], 16: order[6:]}


def long_prefixes(symbols_by_len, lut_bits=10):
    """Distinct lut_bits-bit prefixes among the codes longer than lut_bits."""
    return len({code >> (ln - lut_bits) for code, ln in _canonical(symbols_by_len).values() if ln > lut_bits})


def recode(data, coefs, comps_of_mcu, ac_table_of=long_code_ac_table, dc_tab=None):
    """The file `data` (baseline, one scan, no DRI) with its coefficients `coefs` (n_blocks x 64,
    zigzag order, decode order) coded again: DC tables dc_tab (default: twelve 4-bit codes), AC tables from
    ac_table_of(symbol frequencies).  comps_of_mcu: component index of every block of an MCU,
    e.g. [0, 0, 0, 0, 1, 2].  Table 0 serves component 0, table 1 the others.
    Returns (file, {table id: symbols_by_len of its AC table})."""
    segs, ent = _segments(data)
    assert not any(m == 0xDD for m, _, _ in segs), "no restart intervals here"
    coefs = np.asarray(coefs)
    assert coefs.shape[0] % len(comps_of_mcu) == 0
    ncomp = max(comps_of_mcu) + 1
    pred = [0] * ncomp
    blocks, freq = [], [dict(), dict()]
    for i, blk in enumerate(coefs):
        c = comps_of_mcu[i % len(comps_of_mcu)]
        cat, dbits, ac = _block_symbols(blk, pred[c])
        pred[c] = int(blk[0])
        t = 0 if c == 0 else 1
        for s, _, _ in ac:
            freq[t][s] = freq[t].get(s, 0) + 1
        blocks.append((t, cat, dbits, ac))
    n_tab = 1 if ncomp == 1 else 2
    dc_tab = dc_tab or {4: list(range(12))}
    ac_tabs = {t: ac_table_of(freq[t]) for t in range(n_tab)}
    dc_code = _canonical(dc_tab)
    ac_code = {t: _canonical(ac_tabs[t]) for t in range(n_tab)}
    acc, nbits, out = 0, 0, bytearray()

    def put(v, n):
        nonlocal acc, nbits
        acc = (acc << n) | (v & ((1 << n) - 1))
        nbits += n
        while nbits >= 8:
            byte = (acc >> (nbits - 8)) & 0xFF
            out.append(byte)
            if byte == 0xFF:
                out.append(0)
            nbits -= 8
        acc &= (1 << nbits) - 1

    for t, cat, dbits, ac in blocks:
        put(*dc_code[cat])
        put(dbits, cat)
        for s, size, bits in ac:
            put(*ac_code[t][s])
            put(bits, size)
    if nbits:
        put((1 << (8 - nbits)) - 1, 8 - nbits)
    head = bytearray(b"\xff\xd8")
    for m, a, b in segs:
        if m == 0xC4:
            continue
        if m == 0xDA:
            for t in range(n_tab):
                head += _dht(0, t, dc_tab) + _dht(1, t, ac_tabs[t])
        head += data[a:b]
    return bytes(head) + bytes(out) + b"\xff\xd9", ac_tabs
