"""Test helper: writes a TensorFlow-style object-based checkpoint (tensor bundle + object graph) from a {path: array} mapping,
following the published formats independently of the reader in estts/tf_checkpoint.py (encoders here, decoders there).
Not a general writer: one shard, no compression, no slices."""
import struct

import numpy as np

from estts.tf_checkpoint import TABLE_MAGIC, crc32c, mask_crc

DT = {np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('int64'): 9}


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field_varint(no, v):
    return varint(no << 3) + varint(v)


def field_bytes(no, b):
    return varint((no << 3) | 2) + varint(len(b)) + b


def field_fixed32(no, v):
    return varint((no << 3) | 5) + struct.pack('<I', v)


def table_block(items, restart_interval=16):
    body, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        body += struct.pack('<I', r)
    body += struct.pack('<I', max(len(restarts), 1))
    return bytes(body)


def write_table(path, items, block_bytes=4096):
    """items: sorted [(key bytes, value bytes)]."""
    out, index = bytearray(), []

    def emit(block):
        off = len(out)
        out.extend(block + b'\x00')
        out.extend(struct.pack('<I', mask_crc(crc32c(block + b'\x00'))))
        return varint(off) + varint(len(block))

    group, size = [], 0
    for k, v in items:
        group.append((k, v))
        size += len(k) + len(v)
        if size >= block_bytes:
            index.append((group[-1][0], emit(table_block(group))))
            group, size = [], 0
    if group:
        index.append((group[-1][0], emit(table_block(group))))
    meta = emit(table_block([]))
    idx = emit(table_block(index, restart_interval=1))
    footer = meta + idx
    out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as f:
        f.write(out)


def write_checkpoint(prefix, variables, block_bytes=4096):
    """variables: {object path tuple, e.g. ('net', 'encoder', 'encoder_SADB', '0', 'ffn', 'd1', 'kernel'): ndarray}."""
    nodes = [{'children': {}, 'attributes': []}]
    data, entries = bytearray(), {}

    def add_tensor(key, raw, dtype_code, shape):
        shape_proto = b''.join(field_bytes(2, field_varint(1, d)) for d in shape)
        entries[key] = (field_varint(1, dtype_code) + field_bytes(2, shape_proto) + field_varint(4, len(data)) + field_varint(5, len(raw))
                        + field_fixed32(6, mask_crc(crc32c(raw))))
        data.extend(raw)

    for path, value in variables.items():
        node = 0
        for name in path:
            kids = nodes[node]['children']
            if name not in kids:
                nodes.append({'children': {}, 'attributes': []})
                kids[name] = len(nodes) - 1
            node = kids[name]
        key = '/'.join(path) + '/.ATTRIBUTES/VARIABLE_VALUE'
        nodes[node]['attributes'].append(('VARIABLE_VALUE', '_'.join(path) + ':0', key))
        a = np.asarray(value)
        add_tensor(key, a.astype(a.dtype.newbyteorder('<')).tobytes(), DT[a.dtype], a.shape)
    graph = b''
    for n in nodes:
        body = b''.join(field_bytes(1, field_varint(1, nid) + field_bytes(2, name.encode())) for name, nid in n['children'].items())
        body += b''.join(field_bytes(2, field_bytes(1, a.encode()) + field_bytes(2, full.encode()) + field_bytes(3, ck.encode()))
                         for a, full, ck in n['attributes'])
        graph += field_bytes(1, body)
    lengths = varint(len(graph))
    add_tensor('_CHECKPOINTABLE_OBJECT_GRAPH', lengths + struct.pack('<I', mask_crc(crc32c(lengths))) + graph, 7, ())
    header = field_varint(1, 1) + field_varint(2, 0) + field_bytes(3, field_varint(1, 1))
    items = [(b'', header)] + sorted((k.encode(), v) for k, v in entries.items())
    write_table(prefix + '.index', items, block_bytes)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(data)
