"""Read-only LMDB environment in pure Python (this image has no `lmdb` package / liblmdb), plus a small writer used to
build test routes and synthetic datasets.

The reference opens every recorded route as an LMDB environment and only ever calls `txn.get(key)`
(lav/utils/datasets/basic_dataset.py:47-55, 80-99).  What is restated here is the ON-DISK FORMAT of LMDB 0.9
(`data.mdb`, format version 1 - liblmdb's mdb.c: MDB_meta / MDB_page / MDB_node), the subset a plain key -> value
database uses: two meta pages, branch and leaf pages of a B+tree with byte-wise ordered keys, overflow pages for large
values.  Sub-databases, duplicate-sorted keys and LEAF2 pages do not occur in these files and are refused.

PARITY UNPINNED for the file format: no liblmdb exists in this image to write or read a file with, so the reader is checked
against files from this module's own writer only (tests/test_data_host.py: tree depths 1-3, overflow values, every key
found, absent keys rejected).  NO FILE WRITTEN BY liblmdb / py-lmdb HAS EVER BEEN READ BY THIS MODULE (a search of the image
found no .mdb file and no lmdb or cv2 package to produce a fixture with): treat the recorded-route path of the trainers as
experimental until one has.  The surface mirrors py-lmdb's: open(path, ...).begin(write=False).get(key).
"""
from __future__ import annotations

import builtins
import mmap
import os
import struct
from typing import Iterable, Iterator, Optional, Tuple

MAGIC = 0xBEEFC0DE
VERSION = 1
PAGEHDR = 16
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
P_INVALID = 0xFFFFFFFFFFFFFFFF
_META = struct.Struct("<IIQQ")            # magic, version, address, mapsize
_DB = struct.Struct("<IHHQQQQQ")          # pad (page size in the free DB's record), flags, depth, branch, leaf, overflow pages, entries, root
_NODE = struct.Struct("<HHHH")            # lo, hi, flags, ksize


class Error(Exception):
    pass


class Transaction:
    """A read snapshot: the B+tree under the newer of the two meta pages at the time the environment was opened."""

    def __init__(self, env: "Environment"):
        self._env = env

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def abort(self):
        pass

    commit = abort

    # -- page helpers -----------------------------------------------------------------------------------------
    def _page(self, pgno: int) -> Tuple[int, int, int, int]:
        """(offset, flags, lower, upper) of a page."""
        e = self._env
        off = pgno * e.psize
        if pgno >= e.npages:
            raise Error(f"page {pgno} beyond the end of the file")
        _, _, flags, lower, upper = struct.unpack_from("<QHHHH", e.buf, off)
        return off, flags, lower, upper

    def _node(self, page_off: int, i: int):
        """(node offset, lo, hi, flags, key) of node i of a page."""
        buf = self._env.buf
        ptr, = struct.unpack_from("<H", buf, page_off + PAGEHDR + 2 * i)
        no = page_off + ptr
        lo, hi, flags, ksize = _NODE.unpack_from(buf, no)
        return no, lo, hi, flags, bytes(buf[no + 8:no + 8 + ksize])

    def _leaf_for(self, key: bytes) -> Optional[int]:
        e = self._env
        if e.root == P_INVALID:
            return None
        pgno = e.root
        for _ in range(64):
            off, flags, lower, _ = self._page(pgno)
            if flags & P_LEAF2:
                raise Error("LEAF2 pages (fixed-size duplicate keys) are not supported")
            if flags & P_LEAF:
                return off
            if not flags & P_BRANCH:
                raise Error(f"page {pgno} is neither a branch nor a leaf (flags {flags:#x})")
            n = (lower - PAGEHDR) >> 1
            # last node whose key <= key; node 0 carries no key (it covers everything below node 1's key)
            lo_i, hi_i = 1, n - 1
            pick = 0
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) >> 1
                if self._node(off, mid)[4] <= key:
                    pick, lo_i = mid, mid + 1
                else:
                    hi_i = mid - 1
            _, lo, hi, nflags, _ = self._node(off, pick)
            pgno = lo | (hi << 16) | (nflags << 32)
        raise Error("tree deeper than 64 levels: corrupt file")

    def _value(self, no: int, lo: int, hi: int, flags: int, ksize: int) -> bytes:
        e = self._env
        if flags & (F_SUBDATA | F_DUPDATA):
            raise Error("sub-databases / duplicate-sorted values are not supported")
        size = lo | (hi << 16)
        d = no + 8 + ksize
        if flags & F_BIGDATA:
            pgno, = struct.unpack_from("<Q", e.buf, d)
            off, pflags, _, _ = self._page(pgno)
            if not pflags & P_OVERFLOW:
                raise Error(f"page {pgno} should be an overflow page")
            return bytes(e.buf[off + PAGEHDR:off + PAGEHDR + size])
        return bytes(e.buf[d:d + size])

    # -- py-lmdb surface ---------------------------------------------------------------------------------------
    def get(self, key: bytes, default=None):
        off = self._leaf_for(bytes(key))
        if off is None:
            return default
        lower, = struct.unpack_from("<H", self._env.buf, off + 12)
        n = (lower - PAGEHDR) >> 1
        lo_i, hi_i = 0, n - 1
        while lo_i <= hi_i:
            mid = (lo_i + hi_i) >> 1
            no, lo, hi, flags, k = self._node(off, mid)
            if k == key:
                return self._value(no, lo, hi, flags, len(k))
            if k < key:
                lo_i = mid + 1
            else:
                hi_i = mid - 1
        return default

    def items(self) -> Iterator[Tuple[bytes, bytes]]:
        """All (key, value) pairs in key order (depth-first walk)."""
        e = self._env
        if e.root == P_INVALID:
            return
        stack = [e.root]
        while stack:
            off, flags, lower, _ = self._page(stack.pop())
            n = (lower - PAGEHDR) >> 1
            if flags & P_LEAF:
                for i in range(n):
                    no, lo, hi, nflags, k = self._node(off, i)
                    yield k, self._value(no, lo, hi, nflags, len(k))
            else:
                kids = []
                for i in range(n):
                    _, lo, hi, nflags, _ = self._node(off, i)
                    kids.append(lo | (hi << 16) | (nflags << 32))
                stack.extend(reversed(kids))


class Environment:
    def __init__(self, path: str, subdir: bool = True, **_ignored):
        fname = os.path.join(path, "data.mdb") if subdir and os.path.isdir(path) else path
        self.path = path
        self._f = builtins.open(fname, "rb")
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            raise Error(f"{fname}: too small to be an LMDB data file")
        self.buf = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        best = None
        # the page size is only known from the meta page itself: meta 0 is at offset 0, meta 1 one page further
        m0 = self._read_meta(0)
        psize = m0[0]
        for m in (m0, self._read_meta(psize)):
            if best is None or m[2] > best[2]:
                best = m
        self.psize, self.main, self.txnid, self.last_pg = best
        self.npages = size // self.psize
        flags, self.depth, self.entries, self.root = self.main[1], self.main[2], self.main[6], self.main[7]
        if flags & ~0x08:    # MDB_REVERSEKEY 0x02, DUPSORT 0x04, INTEGERKEY 0x08 ... none of which these files use
            raise Error(f"main database flags {flags:#x} are not supported")

    def _read_meta(self, off: int):
        _, _, flags = struct.unpack_from("<QHH", self.buf, off)
        if not flags & P_META:
            raise Error(f"page at {off} is not a meta page")
        magic, version, _, _ = _META.unpack_from(self.buf, off + PAGEHDR)
        if magic != MAGIC:
            raise Error(f"bad magic {magic:#x}")
        if version != VERSION:
            raise Error(f"data format version {version} is not 1")
        free = _DB.unpack_from(self.buf, off + PAGEHDR + _META.size)
        main = _DB.unpack_from(self.buf, off + PAGEHDR + _META.size + _DB.size)
        last_pg, txnid = struct.unpack_from("<QQ", self.buf, off + PAGEHDR + _META.size + 2 * _DB.size)
        return free[0], main, txnid, last_pg

    def begin(self, write: bool = False, **_ignored) -> Transaction:
        if write:
            raise Error("this environment is read-only")
        return Transaction(self)

    def stat(self):
        return dict(psize=self.psize, depth=self.depth, entries=self.entries)

    def close(self):
        try:
            self.buf.close()
        finally:
            self._f.close()


def open(path: str, **kwargs) -> Environment:   # noqa: A001  (py-lmdb's name)
    """py-lmdb's `lmdb.open(path, max_readers=1, readonly=True, lock=False, readahead=False, meminit=False)`."""
    if kwargs.get("readonly") is False:
        raise Error("this environment is read-only")
    return Environment(path, subdir=kwargs.get("subdir", True))


# ------------------------------------------------------------------------------------------------------------ writer
def write(path: str, items: Iterable[Tuple[bytes, bytes]], psize: int = 4096) -> None:
    """Write `items` as a fresh single-database LMDB environment (directory `path` with data.mdb), bulk-loaded: leaves filled
    left to right in key order, branch levels built bottom-up.  For test routes and synthetic datasets."""
    items = sorted((bytes(k), bytes(v)) for k, v in items)
    for (a, _), (b, _) in zip(items, items[1:]):
        if a == b:
            raise Error(f"duplicate key {a!r}")
    nodemax = (((psize - PAGEHDR) // 2) & -2) - 2
    maxkey = 511
    pages = [None, None]                     # pgno -> bytes; 0 and 1 are the meta pages
    counts = dict(branch=0, leaf=0, overflow=0)

    def new_page(data: bytes) -> int:
        pages.append(data)
        return len(pages) - 1

    def build_page(flags: int, nodes) -> bytes:
        """nodes: list of raw node byte strings; laid out from the end of the page downwards, as liblmdb does."""
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for nd in nodes:
            sz = (len(nd) + 1) & ~1
            upper -= sz
            buf[upper:upper + len(nd)] = nd
            ptrs.append(upper)
        lower = PAGEHDR + 2 * len(nodes)
        assert lower <= upper, "page overflow"
        struct.pack_into("<QHHHH", buf, 0, 0, 0, flags, lower, upper)
        for i, p_ in enumerate(ptrs):
            struct.pack_into("<H", buf, PAGEHDR + 2 * i, p_)
        return bytes(buf)

    def flush(flags, nodes, first_key, level):
        pg = new_page(build_page(flags, nodes))
        counts["leaf" if flags & P_LEAF else "branch"] += 1
        level.append((first_key, pg))

    # leaves
    level = []
    nodes, used, first = [], PAGEHDR, None
    for k, v in items:
        if not 0 < len(k) <= maxkey:
            raise Error(f"key of {len(k)} bytes (1..{maxkey} allowed)")
        if 8 + len(k) + len(v) > nodemax:                       # value goes to overflow pages
            npg = (PAGEHDR + len(v) + psize - 1) // psize
            body = bytearray(npg * psize)
            struct.pack_into("<QHHI", body, 0, 0, 0, P_OVERFLOW, npg)
            body[PAGEHDR:PAGEHDR + len(v)] = v
            first_pg = len(pages)
            for i in range(npg):
                pages.append(bytes(body[i * psize:(i + 1) * psize]))
            counts["overflow"] += npg
            nd = _NODE.pack(len(v) & 0xFFFF, len(v) >> 16, F_BIGDATA, len(k)) + k + struct.pack("<Q", first_pg)
        else:
            nd = _NODE.pack(len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        sz = ((len(nd) + 1) & ~1) + 2
        if nodes and used + sz > psize:
            flush(P_LEAF, nodes, first, level)
            nodes, used, first = [], PAGEHDR, None
        if first is None:
            first = k
        nodes.append(nd)
        used += sz
    if nodes:
        flush(P_LEAF, nodes, first, level)
    depth = 1 if level else 0
    # branch levels
    while len(level) > 1:
        upper_level = []
        nodes, used, first = [], PAGEHDR, None
        for k, pg in level:
            key = b"" if not nodes else k                       # a branch page's first node has no key
            nd = _NODE.pack(pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(key)) + key
            sz = ((len(nd) + 1) & ~1) + 2
            if len(nodes) >= 2 and used + sz > psize:
                flush(P_BRANCH, nodes, first, upper_level)
                nodes, used, first = [], PAGEHDR, None
                nd = _NODE.pack(pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
                sz = 8 + 2
            if first is None:
                first = k
            nodes.append(nd)
            used += sz
        flush(P_BRANCH, nodes, first, upper_level)
        level = upper_level
        depth += 1
    root = level[0][1] if level else P_INVALID
    last_pg = len(pages) - 1

    def meta(pgno: int, txnid: int, live: bool) -> bytes:
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, P_META, 0, 0)
        o = PAGEHDR
        _META.pack_into(buf, o, MAGIC, VERSION, 0, max(len(pages) * psize, 1 << 20))
        o += _META.size
        _DB.pack_into(buf, o, psize, 0x08, 0, 0, 0, 0, 0, P_INVALID)                        # free-page database: empty, integer keys
        o += _DB.size
        if live:
            _DB.pack_into(buf, o, 0, 0, depth, counts["branch"], counts["leaf"], counts["overflow"], len(items), root)
        else:
            _DB.pack_into(buf, o, 0, 0, 0, 0, 0, 0, 0, P_INVALID)
        o += _DB.size
        struct.pack_into("<QQ", buf, o, last_pg if live else 1, txnid)
        return bytes(buf)

    pages[0] = meta(0, 0, False)
    pages[1] = meta(1, 1, True)
    os.makedirs(path, exist_ok=True)
    with builtins.open(os.path.join(path, "data.mdb"), "wb") as f:
        for p_ in pages:
            f.write(p_)
