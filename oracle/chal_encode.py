"""oracle/chal_encode.py -- TEST INFRASTRUCTURE ONLY (never imported by the product).

An independent Python `struct` writer of the HierarchicalNSWSlim index file format
(/root/reference/third_party/hnswlib/hnswalg_slim.h:717-751, read back by :753-815), used to

  * cross-pin the Slim class to outputs of the COMPILED reference: `vanilla_to_slim_verbatim` re-encodes a vanilla
    hnswlib index file (hnswalg.h:748-779; the golden ones are written by the compiled reference itself) as a Slim file
    with every level's neighbour list kept verbatim and in the reference's order (threshold_level = 0, nothing pruned).
    On such a file HierarchicalNSWSlim::searchKnn(q,k,tableint*) (hnswalg_slim.h:2030-2131) performs the same greedy
    descent and the same level-0 beam as HierarchicalNSW::searchKnn (hnswalg.h:1378-1440), so its k-set, fp32 distances
    and distance-evaluation count (minus the entry distance vanilla recomputes, hnswalg.h:347-351) must equal the
    compiled reference's golden outputs;
  * feed the product's and the oracle's Slim loaders a file that neither of them wrote (stale pointer bytes filled
    with garbage, blobs present iff blobSize != 0 and total_neighbor != 0).

Delete marks cannot be carried over: the Slim class reads the mark from byte 2 of total_neighbor (hnswalg_slim.h:1776-1781)
while get_total_neighbor (:644) reads all four bytes, and convertFromHNSW (:1088) never sets it -- no valid Slim file holds a
marked element.  Marked vanilla fixtures are therefore encoded with has_deleted_elements = 1 and unmarked elements.
"""
import struct

import numpy as np

VANILLA_HDR = "<6QiI3QdQ"   # offsetLevel0, max_elements, count, size_per_el, label_offset, offsetData, maxlevel, ep, maxM, maxM0, M, mult, efC
SLIM_HDR = "<6Q2iI4Q?"      # count, size_per_el, label_offset, offsetTotalNeighbor, offsetData, offsetNeighbor, maxlevel, threshold_level,
                            # enterpoint, maxM, maxM0, M, efC, has_deleted_elements   (93 bytes)


def parse_vanilla(raw):
    """Independent parse of a vanilla index file -> dict(header fields, per-node lists per level, labels, rows, marks)."""
    (off0, max_el, count, spe, label_off, off_data, maxlevel, ep, maxM, maxM0, M, mult, efC) = struct.unpack_from(VANILLA_HDR, raw, 0)
    pos = struct.calcsize(VANILLA_HDR)
    dim = (spe - 4 - 4 * maxM0 - 8) // 4
    assert spe == 4 + 4 * maxM0 + 4 * dim + 8
    l0 = np.frombuffer(raw, np.uint8, count * spe, pos).reshape(count, spe)
    pos += count * spe
    cnt0 = l0[:, off0:off0 + 2].copy().view(np.uint16)[:, 0].astype(np.int64)
    marks = (l0[:, off0 + 2] & 1).astype(np.uint8)
    ids0 = l0[:, off0 + 4:off0 + 4 + 4 * maxM0].copy().view(np.uint32)
    rows = l0[:, off_data:off_data + 4 * dim].copy().view(np.float32)
    labels = l0[:, label_off:label_off + 8].copy().view(np.uint64)[:, 0]
    per = 4 + 4 * maxM
    lists = []
    for i in range(count):
        (sz,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        assert sz % per == 0
        node = [ids0[i, :cnt0[i]].copy()]
        for lv in range(sz // per):
            (c,) = struct.unpack_from("<H", raw, pos + lv * per)
            node.append(np.frombuffer(raw, np.uint32, c, pos + lv * per + 4).copy())
        pos += sz
        lists.append(node)
    assert pos == len(raw)
    return dict(count=count, dim=dim, maxlevel=maxlevel, enterpoint=ep, maxM=maxM, maxM0=maxM0, M=M, efC=efC, lists=lists, labels=labels,
                rows=rows, marks=marks)


def write_slim(g, threshold_level=0, has_deleted=None, garbage_seed=12345):
    """Serialise dict(count, dim, maxlevel, enterpoint, maxM, maxM0, M, efC, lists[node][level] -> ids, labels, rows) as a Slim file."""
    n, dim = g["count"], g["dim"]
    spe = 24 + 4 * dim
    if has_deleted is None:
        has_deleted = bool(np.any(g.get("marks", np.zeros(1, np.uint8))))
    out = [struct.pack(SLIM_HDR, n, spe, 8, 4, 24, 16, g["maxlevel"], threshold_level, g["enterpoint"], g["maxM"], g["maxM0"], g["M"],
                       g["efC"], has_deleted)]
    rng = np.random.default_rng(garbage_seed)
    el = np.zeros((n, spe), np.uint8)
    blobs = []
    for i in range(n):
        node = g["lists"][i]
        level = len(node) - 1
        total = int(sum(len(x) for x in node))
        cum = np.cumsum([len(x) for x in node])[:level].astype(np.uint16)   # off[l] = end of the level-l slice, l < level
        blob = cum.tobytes() + (np.concatenate(node).astype(np.uint32).tobytes() if total else b"")
        assert len(blob) == 2 * level + 4 * total
        blobs.append((blob, total))
        el[i, 0:4] = np.frombuffer(struct.pack("<i", level), np.uint8)
        el[i, 4:8] = np.frombuffer(struct.pack("<I", total), np.uint8)
        el[i, 8:16] = np.frombuffer(struct.pack("<Q", int(g["labels"][i])), np.uint8)
    el[:, 16:24] = rng.integers(1, 256, size=(n, 8), dtype=np.uint8)   # the 8 stale char* bytes saveIndex dumps: never zero here
    el[:, 24:] = np.ascontiguousarray(g["rows"], np.float32).view(np.uint8).reshape(n, 4 * dim)
    out.append(el.tobytes())
    for blob, total in blobs:
        out.append(struct.pack("<I", len(blob)))
        if len(blob) and total:          # hnswalg_slim.h:745-748
            out.append(blob)
    return b"".join(out)


def vanilla_to_slim_verbatim(raw, **kw):
    return write_slim(parse_vanilla(raw), **kw)


def parse_slim(raw, dim):
    """Independent reader of a Slim file (consumes it to the last byte) -> per-node level, lists per level, labels, rows."""
    hdr = struct.unpack_from(SLIM_HDR, raw, 0)
    n, spe = hdr[0], hdr[1]
    assert spe == 24 + 4 * dim and hdr[2:6] == (8, 4, 24, 16)
    pos = struct.calcsize(SLIM_HDR)
    el = np.frombuffer(raw, np.uint8, n * spe, pos).reshape(n, spe)
    pos += n * spe
    level = el[:, 0:4].copy().view(np.int32)[:, 0]
    total = el[:, 4:8].copy().view(np.uint32)[:, 0]
    lists = []
    for i in range(n):
        (sz,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        L, T = int(level[i]), int(total[i])
        assert sz == 2 * L + 4 * T
        if sz == 0 or T == 0:
            lists.append([np.zeros(0, np.uint32) for _ in range(L + 1)])
            continue
        off = np.frombuffer(raw, np.uint16, L, pos).astype(np.int64)
        ids = np.frombuffer(raw, np.uint32, T, pos + 2 * L)
        pos += sz
        bounds = [0] + list(off) + [T]
        lists.append([ids[bounds[l]:bounds[l + 1]].copy() for l in range(L + 1)])
    assert pos == len(raw)
    return dict(count=n, dim=dim, maxlevel=hdr[6], threshold_level=hdr[7], enterpoint=hdr[8], maxM=hdr[9], maxM0=hdr[10], M=hdr[11],
                efC=hdr[12], has_deleted=hdr[13], level=level, lists=lists, labels=el[:, 8:16].copy().view(np.uint64)[:, 0],
                rows=el[:, 24:].copy().view(np.float32))


# ---- HierarchicalNSWSlimQ file (hnswalg_slimq.h:1161-1216 / 1218-1313) -------------------------------------------------
SLIMQ_EXT = "<9QB"   # num_cluster, dim, padded_dim, offset_cluster_id, offset_bin_data, offset_ex_data, size_bin_data, size_ex_data, ex_bits, metric


def parse_slimq(raw):
    """Independent reader of a SlimQ file: Slim header + quantiser header + rotated centroids + rotator flips + elements
    [i32 level][u32 total][u64 label][8 B ptr][u32 cluster][code padded/8 B][f_add f_rescale f_error][ex area] + blobs."""
    h = struct.unpack_from(SLIM_HDR, raw, 0)
    pos = struct.calcsize(SLIM_HDR)
    x = struct.unpack_from(SLIMQ_EXT, raw, pos)
    pos += struct.calcsize(SLIMQ_EXT)
    n, spe = h[0], h[1]
    ncl, dim, padded, off_cid, off_bin, off_ex, sbin, sex, ex_bits, metric = x
    assert padded % 64 == 0 and sbin == padded // 8 + 12 and off_cid == 24 and off_bin == 28 and off_ex == 28 + sbin and spe == 28 + sbin + sex
    cent = np.frombuffer(raw, np.float32, ncl * padded, pos).reshape(ncl, padded).copy()
    pos += ncl * padded * 4
    nflip = 4 * padded // 8
    flips = np.frombuffer(raw, np.uint8, nflip, pos).copy()
    pos += nflip
    el = np.frombuffer(raw, np.uint8, n * spe, pos).reshape(n, spe)
    pos += n * spe
    level = el[:, 0:4].copy().view(np.int32)[:, 0]
    total = el[:, 4:8].copy().view(np.uint32)[:, 0]
    blobs = []
    for i in range(n):
        (sz,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        assert sz == 2 * int(level[i]) + 4 * int(total[i])
        if sz and total[i]:
            blobs.append(bytes(raw[pos:pos + sz]))
            pos += sz
        else:
            blobs.append(b"")
    assert pos == len(raw)
    return dict(hdr=h, ext=x, centroids=cent, flips=flips, level=level, total=total, labels=el[:, 8:16].copy().view(np.uint64)[:, 0],
                cluster=el[:, 24:28].copy().view(np.uint32)[:, 0], code=el[:, 28:28 + padded // 8].copy(),
                factors=el[:, 28 + padded // 8:28 + padded // 8 + 12].copy().view(np.float32), blobs=blobs)


def write_slimq(g, garbage_seed=777):
    """Re-serialise a parsed SlimQ file from its fields; the bytes no function of the search path reads -- the 8 stale pointer bytes
    of every element and the whole ex-data area (hnswalg_slimq.h:1498-1505; searchKnn(q,k,result) only touches the 1-bit
    code and its three factors) -- are filled with garbage."""
    h, x = g["hdr"], g["ext"]
    n, spe = h[0], h[1]
    padded, sbin = x[2], x[6]
    rng = np.random.default_rng(garbage_seed)
    el = rng.integers(1, 256, size=(n, spe), dtype=np.uint8)
    el[:, 0:4] = g["level"].astype("<i4").view(np.uint8).reshape(n, 4)
    el[:, 4:8] = g["total"].astype("<u4").view(np.uint8).reshape(n, 4)
    el[:, 8:16] = g["labels"].astype("<u8").view(np.uint8).reshape(n, 8)
    el[:, 24:28] = g["cluster"].astype("<u4").view(np.uint8).reshape(n, 4)
    el[:, 28:28 + padded // 8] = g["code"]
    el[:, 28 + padded // 8:28 + sbin] = np.ascontiguousarray(g["factors"], np.float32).view(np.uint8).reshape(n, 12)
    out = [struct.pack(SLIM_HDR, *h), struct.pack(SLIMQ_EXT, *x), np.ascontiguousarray(g["centroids"], np.float32).tobytes(), g["flips"].tobytes(),
           el.tobytes()]
    for i in range(n):
        b = g["blobs"][i]
        out.append(struct.pack("<I", 2 * int(g["level"][i]) + 4 * int(g["total"][i])))
        if b:
            out.append(b)
    return b"".join(out)


# ---- the diff / patch wire format (genPatch hnswalg_slim.h:1427-1476, framed as hnsw_slim_server_patch.cc:280-290 does) ------
def make_patch(raw_old, raw_new, dim, to_add=True):
    """Patch stream that turns Slim file `raw_old` (n nodes) into `raw_new` (n + delta nodes, the first n being the same points):
    u64 cur_element_count, u64 changed_old_cnt, u64 changed_new_cnt, then the changed old nodes (8-byte head) and the new nodes
    (16-byte head, + vector when to_add).  Test infrastructure: the reference derives the changed sets in
    convertFromHNSWWithDiff; here they come from comparing the two files."""
    a, b = parse_slim(raw_old, dim), parse_slim(raw_new, dim)
    n_old, n_new = a["count"], b["count"]
    assert n_new >= n_old

    def blob(g, i):
        node = g["lists"][i]
        level = len(node) - 1
        total = int(sum(len(x) for x in node))
        cum = np.cumsum([len(x) for x in node])[:level].astype(np.uint16)
        return level, total, (cum.tobytes() + (np.concatenate(node).astype(np.uint32).tobytes() if total else b""))

    old_ids = []
    for i in range(n_old):
        la, lb = a["lists"][i], b["lists"][i]
        if len(la) != len(lb) or any(not np.array_equal(x, y) for x, y in zip(la, lb)):
            old_ids.append(i)
    out = [struct.pack("<3Q", n_new, len(old_ids), n_new - n_old)]
    for i in old_ids:
        level, total, bl = blob(b, i)
        out.append(struct.pack("<IiI", i, level, total))
        out.append(struct.pack("<I", len(bl) if total else 0))
        if total:
            out.append(bl)
    for i in range(n_old, n_new):
        level, total, bl = blob(b, i)
        out.append(struct.pack("<IiIQ", i, level, total, int(b["labels"][i])))
        out.append(struct.pack("<I", len(bl) if total else 0))
        if total:
            out.append(bl)
        if to_add:
            out.append(np.ascontiguousarray(b["rows"][i], np.float32).tobytes())
    return b"".join(out), len(old_ids), n_new - n_old


def with_entry_of(raw_new, raw_old):
    """`raw_new` with the enter point and max level of `raw_old` (the patch stream does not carry them: hnswalg_slim.h:2292-2340)."""
    h_old, h_new = list(struct.unpack_from(SLIM_HDR, raw_old, 0)), list(struct.unpack_from(SLIM_HDR, raw_new, 0))
    h_new[6], h_new[8] = h_old[6], h_old[8]
    return struct.pack(SLIM_HDR, *h_new) + raw_new[struct.calcsize(SLIM_HDR):]
