"""Minimal HDF5 reader for the one file the hot path needs: assets/realistic_arm_limits_model.h5, the Keras model behind
Human.enforce_realistic_joint_limits (assistive_gym/envs/agents/human.py:134-152, loaded at envs/env.py:39).

h5py is not installed here; the file is HDF5 superblock version 0 with old-style groups (symbol table B-trees +
local heaps), version 1 object headers and contiguous, uncompressed little-endian datasets.  Exactly
that subset of the HDF5 file format specification is implemented: enough to walk the group tree and return every
dataset as a numpy array.  Anything else (chunked / compressed layouts, new-style groups, big-endian types) raises.
"""
import struct

import numpy as np

SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class H5File:
    def __init__(self, path):
        self.d = open(path, 'rb').read()
        d = self.d
        if d[:8] != SIG:
            raise H5Error('not an HDF5 file')
        if d[8] != 0:
            raise H5Error('superblock version %d not supported' % d[8])
        self.so, self.sl = d[13], d[14]                     # size of offsets / lengths
        if (self.so, self.sl) != (8, 8):
            raise H5Error('only 8-byte offsets and lengths are supported')
        # superblock v0: ... base address, free-space address, end of file, driver info, then the root symbol table entry
        base = 24
        self.base_addr = self._u64(base)
        root = base + 4 * 8
        self.root_header = self._symbol_entry(root)[1]

    # ---- primitives -------------------------------------------------------------------------------------
    def _u16(self, o):
        return struct.unpack_from('<H', self.d, o)[0]

    def _u32(self, o):
        return struct.unpack_from('<I', self.d, o)[0]

    def _u64(self, o):
        return struct.unpack_from('<Q', self.d, o)[0]

    def _symbol_entry(self, o):
        """(link name offset in the local heap, object header address, cache type, scratch)"""
        return self._u64(o), self._u64(o + 8), self._u32(o + 16), self.d[o + 24:o + 40]

    # ---- object headers (version 1) -------------------------------------------------------------------------
    def _messages(self, addr):
        d = self.d
        if d[addr] != 1:
            raise H5Error('object header version %d not supported' % d[addr])
        nmsg = self._u16(addr + 2)
        size = self._u32(addr + 8)
        out = []
        blocks = [(addr + 16, size)]
        while blocks and len(out) < nmsg:
            o, left = blocks.pop(0)
            end = o + left
            while o + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = self._u16(o), self._u16(o + 2), d[o + 4]
                body = o + 8
                if mtype == 0x0010:                          # continuation: (offset, length) of another block of messages
                    blocks.append((self._u64(body), self._u64(body + 8)))
                out.append((mtype, body, msize))
                o = body + msize
        return out

    def _heap_data(self, addr):
        if self.d[addr:addr + 4] != b'HEAP':
            raise H5Error('bad local heap')
        return self._u64(addr + 24)                          # address of the data segment

    def _group_entries(self, btree, heap):
        """names and object header addresses of a symbol-table group"""
        data = self._heap_data(heap)
        out = []

        def walk(node):
            d = self.d
            if d[node:node + 4] != b'TREE' or d[node + 4] != 0:
                raise H5Error('bad group B-tree node')
            level, used = d[node + 5], self._u16(node + 6)
            o = node + 8 + 16                                # skip the sibling addresses
            for k in range(used):
                child = self._u64(o + 8)                     # key, child, key, child, ..., key
                o += 16
                if level > 0:
                    walk(child)
                else:
                    if d[child:child + 4] != b'SNOD':
                        raise H5Error('bad symbol table node')
                    n = self._u16(child + 6)
                    for e in range(n):
                        name_off, hdr, _, _ = self._symbol_entry(child + 8 + 40 * e)
                        s = data + name_off
                        name = d[s:d.index(b'\0', s)].decode()
                        out.append((name, hdr))
        walk(btree)
        return out

    def _dataset(self, msgs):
        d = self.d
        shape, dtype, addr, nbytes = None, None, None, None
        for mtype, body, msize in msgs:
            if mtype == 0x0001:                              # dataspace
                ver, rank, flags = d[body], d[body + 1], d[body + 2]
                o = body + (8 if ver == 1 else 4)
                shape = tuple(self._u64(o + 8 * k) for k in range(rank))
            elif mtype == 0x0003:                            # datatype
                cls_ver, bits0 = d[body], d[body + 1]
                cls, size = cls_ver & 15, self._u32(body + 4)
                if bits0 & 1:
                    raise H5Error('big-endian data not supported')
                if cls == 1 and size in (4, 8):
                    dtype = np.dtype('<f%d' % size)
                elif cls == 0 and size in (1, 2, 4, 8):
                    dtype = np.dtype('<%s%d' % ('i' if d[body + 1] & 8 else 'u', size))
                else:
                    dtype = None                             # strings etc.: not needed
            elif mtype == 0x0008:                            # data layout
                ver = d[body]
                if ver == 3:
                    if d[body + 1] != 1:
                        raise H5Error('only contiguous datasets are supported (layout class %d)' % d[body + 1])
                    addr, nbytes = self._u64(body + 2), self._u64(body + 10)
                elif ver in (1, 2):
                    rank, cls = d[body + 1], d[body + 2]
                    if cls != 1:
                        raise H5Error('only contiguous datasets are supported')
                    addr = self._u64(body + 8)
                else:
                    raise H5Error('data layout version %d not supported' % ver)
            elif mtype == 0x000B:
                raise H5Error('filtered (compressed) datasets are not supported')
        if shape is None or dtype is None or addr is None or addr == UNDEF:
            return None
        n = int(np.prod(shape)) if shape else 1
        return np.frombuffer(d, dtype=dtype, count=n, offset=addr + self.base_addr).reshape(shape).copy()

    # ---- public -----------------------------------------------------------------------------------------------
    def datasets(self):
        """{'/group/.../name': ndarray} of every numeric dataset in the file"""
        out = {}

        def visit(path, hdr, depth=0):
            msgs = self._messages(hdr)
            sym = [m for m in msgs if m[0] == 0x0011]
            if sym:                                          # a group: B-tree + local heap
                body = sym[0][1]
                for name, child in self._group_entries(self._u64(body), self._u64(body + 8)):
                    if depth < 16:
                        visit(path + '/' + name, child, depth + 1)
            else:
                try:
                    arr = self._dataset(msgs)
                except H5Error:
                    raise
                if arr is not None:
                    out[path] = arr
        visit('', self.root_header)
        return out


def load_keras_dense_stack(path):
    """[(kernel [in, out], bias [out]), ...] of a Keras Sequential of Dense layers, in layer order (dense_1, dense_2, ...)"""
    ds = H5File(path).datasets()
    layers = {}
    for name, arr in ds.items():
        parts = name.strip('/').split('/')
        if parts[0] != 'model_weights' or len(parts) < 3:
            continue
        layer, leaf = parts[1], parts[-1]
        if leaf.startswith('kernel'):
            layers.setdefault(layer, {})['kernel'] = arr.astype(np.float32)
        elif leaf.startswith('bias'):
            layers.setdefault(layer, {})['bias'] = arr.astype(np.float32)
    names = sorted(layers, key=lambda n: int(n.split('_')[-1]))
    return [(layers[n]['kernel'], layers[n]['bias']) for n in names]
