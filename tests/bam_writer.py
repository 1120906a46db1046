"""Test-side writers for BAM (+BAI) and FASTA (+FAI), implemented from the SAM/BAM specification with the
standard library only, plus a Python restatement of BAM_handler.get_reads' region clipping
(bam_handler.cpp:180-306) used to cross-check the native reader."""
import struct
import zlib

import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
NT16_CODE = {c: i for i, c in enumerate(NT16)}


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def ref_len(cigar):
    return sum(l for op, l in cigar if op in (0, 2, 3, 7, 8))


def write_bam(path, refs, records, block_records=40):
    """refs: [(name, length)]; records: dicts(tid,pos,mapq,flag,cigar[(op,len)],seq(str),qual(list),name,hp)
    sorted by (tid,pos). Writes path and path+'.bai'."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    hdr = b"BAM\x01" + struct.pack("<I", len(text)) + text.encode() + struct.pack("<I", len(refs))
    for name, ln in refs:
        hdr += struct.pack("<I", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<I", ln)
    out = bytearray()
    out += _bgzf_block(hdr)
    index = [dict(bins={}, linear={}) for _ in refs]
    cur = bytearray()

    def flush():
        nonlocal cur
        if cur:
            out.extend(_bgzf_block(bytes(cur)))
            cur = bytearray()

    n_in_block = 0
    for rec in records:
        if n_in_block >= block_records:
            flush()
            n_in_block = 0
        cig = rec["cigar"]
        seq = rec["seq"]
        end = rec["pos"] + max(ref_len(cig), 1)
        b = reg2bin(rec["pos"], end)
        name = rec.get("name", "r").encode() + b"\x00"
        packed = bytearray((len(seq) + 1) // 2)
        for i, c in enumerate(seq):
            packed[i >> 1] |= NT16_CODE[c] << (4 if i % 2 == 0 else 0)
        aux = b""
        if rec.get("hp") is not None:
            aux += b"HPC" + struct.pack("<B", rec["hp"])
        aux += b"NMi" + struct.pack("<i", 3) + b"RGZgrp1\x00"
        cig_rec = cig
        if rec.get("cg"):
            # SAMv1 4.2.2: a CIGAR too long for the 16-bit count lives in the CG:B,I tag behind a <l_seq>S<rlen>N placeholder
            cig_rec = [(4, len(seq)), (3, ref_len(cig))]
            aux += b"CGBI" + struct.pack("<I", len(cig)) + b"".join(struct.pack("<I", (l << 4) | op) for op, l in cig)
        body = struct.pack("<iiBBHHHIiii", rec["tid"], rec["pos"], len(name), rec["mapq"], b, len(cig_rec), rec["flag"], len(seq), -1, -1, 0)
        body += name + b"".join(struct.pack("<I", (l << 4) | op) for op, l in cig_rec) + bytes(packed) + bytes(rec["qual"]) + aux
        data = struct.pack("<I", len(body)) + body
        if cur and len(cur) + len(data) > 0xFF00:  # a BGZF block holds at most 64 KiB: start the record in a fresh block
            flush()
            n_in_block = 0
        vbeg = (len(out) << 16) | len(cur)
        cur += data
        while len(cur) > 0xFF00:                   # a record longer than a block spans blocks
            head, rest = bytes(cur[:0xFF00]), cur[0xFF00:]
            out.extend(_bgzf_block(head))
            cur = bytearray(rest)
        n_in_block += 1
        vend = (len(out) << 16) | len(cur)
        ix = index[rec["tid"]]
        ix["bins"].setdefault(b, []).append([vbeg, vend])
        for w in range(rec["pos"] >> 14, ((end - 1) >> 14) + 1):
            if w not in ix["linear"] or vbeg < ix["linear"][w]:
                ix["linear"][w] = vbeg
    flush()
    out += BGZF_EOF
    with open(path, "wb") as f:
        f.write(out)
    bai = bytearray(b"BAI\x01" + struct.pack("<I", len(refs)))
    for ix in index:
        bai += struct.pack("<I", len(ix["bins"]))
        for b, chunks in sorted(ix["bins"].items()):
            merged = []
            for c in chunks:
                if merged and c[0] <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], c[1])
                else:
                    merged.append(list(c))
            bai += struct.pack("<II", b, len(merged))
            for c in merged:
                bai += struct.pack("<QQ", c[0], c[1])
        n_intv = (max(ix["linear"]) + 1) if ix["linear"] else 0
        bai += struct.pack("<I", n_intv)
        last = 0
        for w in range(n_intv):
            last = ix["linear"].get(w, last)
            bai += struct.pack("<Q", last)
    with open(path + ".bai", "wb") as f:
        f.write(bai)


def write_fasta(path, seqs, width=60):
    """seqs: [(name, sequence str)]; writes path and path + '.fai'"""
    fai = []
    with open(path, "wb") as f:
        for name, s in seqs:
            f.write((">%s some description\n" % name).encode())
            off = f.tell()
            for i in range(0, len(s), width):
                f.write(s[i:i + width].encode() + b"\n")
            fai.append("%s\t%d\t%d\t%d\t%d\n" % (name, len(s), off, width, width + 1))
    with open(path + ".fai", "w") as f:
        f.writelines(fai)


def clip_read(rec, start, stop):
    """bam_handler.cpp:180-306 in Python: -> (pos_start, pos_end, seq, quals, cigar) or None"""
    pos_start = pos_end = -1
    cur_pos, cur_idx = rec["pos"], 0
    seq, quals, cig = [], [], []
    for op, ln in rec["cigar"]:
        if cur_pos > stop:
            break
        kept = 0
        if op in (0, 7, 8):
            i0 = 0
            if cur_pos < start:
                i0 = min(start - cur_pos, ln)
                cur_idx += i0
                cur_pos += i0
            for _ in range(i0, ln):
                if cur_pos <= stop:
                    if pos_start == -1:
                        pos_start = pos_end = cur_pos
                    seq.append(rec["seq"][cur_idx].upper())
                    quals.append(rec["qual"][cur_idx])
                    kept += 1
                    pos_end += 1
                else:
                    break
                cur_idx += 1
                cur_pos += 1
        elif op in (4, 1):
            if start <= cur_pos <= stop and pos_start != -1:
                for _ in range(ln):
                    seq.append(rec["seq"][cur_idx].upper())
                    quals.append(rec["qual"][cur_idx])
                    kept += 1
                    cur_idx += 1
            else:
                cur_idx += ln
        elif op in (3, 2):
            if start <= cur_pos <= stop and pos_start != -1:
                for _ in range(ln):
                    if cur_pos <= stop:
                        kept += 1
                        pos_end += 1
                    else:
                        break
                    cur_pos += 1
            else:
                cur_pos += ln
        if kept > 0:
            cig.append((op, kept))
    if not seq:
        return None
    return pos_start, pos_end, "".join(seq), quals, cig


def expected_reads(records, tid, start, stop, include_supp=False, min_mapq=0):
    """the whole get_reads contract on in-memory records"""
    out = []
    for rec in records:
        if rec["tid"] != tid:
            continue
        end = rec["pos"] + max(ref_len(rec["cigar"]), 1)
        if not (end > max(start, 0) and stop > rec["pos"]):
            continue
        fl = rec["flag"]
        if fl & (0x200 | 0x400 | 0x100 | 0x4):
            continue
        if not include_supp and fl & 0x800:
            continue
        if rec["mapq"] < min_mapq:
            continue
        c = clip_read(rec, start, stop)
        if c is None:
            continue
        out.append(dict(pos=c[0], pos_end=c[1], seq=c[2], qual=c[3], cigar=c[4], rev=bool(fl & 0x10), mapq=rec["mapq"],
                        hp=rec.get("hp") or 0, name=rec.get("name", "r")))
    return out


def random_records(rng, n, ref_len_, tid=0, mean_len=3000, allow_skip=True):
    recs = []
    for i in range(n):
        pos = int(rng.integers(0, max(1, ref_len_ - 200)))
        target = int(max(60, rng.normal(mean_len, mean_len * 0.3)))
        cig, qn, rn = [], 0, 0
        if rng.random() < 0.3:
            s = int(rng.integers(1, 30)); cig.append((4, s)); qn += s
        elif rng.random() < 0.1:
            cig.append((5, int(rng.integers(1, 30))))
        while rn < target and pos + rn < ref_len_ - 50:
            m = int(rng.integers(1, 60)); cig.append((int(rng.choice([0, 0, 0, 7, 8])), m)); qn += m; rn += m
            r = rng.random()
            if r < 0.25:
                k = int(rng.integers(1, 6)); cig.append((1, k)); qn += k
            elif r < 0.5:
                k = int(rng.integers(1, 6)); cig.append((2, k)); rn += k
            elif r < 0.52 and allow_skip:
                k = int(rng.integers(5, 40)); cig.append((3, k)); rn += k
            elif r < 0.54 and allow_skip:
                cig.append((6, int(rng.integers(1, 4))))
        if cig[-1][0] not in (0, 7, 8):
            cig.append((0, 5)); qn += 5; rn += 5
        if rng.random() < 0.3:
            s = int(rng.integers(1, 30)); cig.append((4, s)); qn += s
        seq = "".join(rng.choice(list("ACGTN"), size=qn, p=[.245, .245, .245, .245, .02]))
        flag = int(rng.choice([0, 16, 0, 16, 0x800, 0x810, 0x100, 0x400, 0x200, 4], p=[.35, .35, .08, .08, .04, .03, .03, .02, .01, .01]))
        recs.append(dict(tid=tid, pos=pos, mapq=int(rng.choice([60, 60, 30, 10, 4, 0])), flag=flag, cigar=cig, seq=seq,
                         qual=[int(q) for q in rng.integers(0, 60, size=qn)], name="read%d" % i,
                         hp=(int(rng.integers(1, 3)) if rng.random() < 0.5 else None)))
    recs.sort(key=lambda r: (r["tid"], r["pos"]))
    return recs
