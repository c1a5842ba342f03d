"""TensorFlow object-based checkpoints -> this package's weight dictionaries, without TensorFlow.

The reference saves and restores TransformerTTS weights with `tf.train.Checkpoint(net=model)` + `CheckpointManager`
(TransformerTTS/utils/config_manager.py:203-216; the training scripts add `step` and `optimizer` beside `net`).  Such a
checkpoint is a tensor bundle: `<prefix>.index`, a sorted string table (the LevelDB table format: prefix-compressed blocks with
restart arrays, an index block, a 48-byte footer ending in the magic 0xdb4775248b80fb57) mapping tensor keys to
BundleEntryProto records (dtype, shape, shard, offset, size, crc32c), and `<prefix>.data-SSSSS-of-NNNNN` holding the raw
little-endian tensor bytes.  The key `_CHECKPOINTABLE_OBJECT_GRAPH` holds a serialized TrackableObjectGraph: one node per
tracked Python object with its children by attribute name (list elements by index) and, for variables, the key of their
value.  Variables are located by walking that graph along the reference's attribute names (`net` -> `encoder` ->
`encoder_SADB` -> `0` -> `sarn` -> `mha` -> `wq` -> `kernel`), which does not depend on how TensorFlow spells the keys.

The attribute paths come from the reference's constructors: pretrained_models/for_text_encoder/models.py:315-358
(ForwardTransformer), model/AutoregressiveTransformer.py:74-119, model/layers.py:19-31,58-63,112-116,158-161,180-181,199-200,
229-236,266-268,286-288,311-313,347-354,387-392,538-552,597-598,628-636.

STATUS: written from the published formats; no checkpoint ships with the reference and TensorFlow is not installable here, so
this reader has only been exercised on bundles produced by this repository's own test writer (tests/tf_bundle_writer.py).
`python -m estts.tf_checkpoint <prefix> --list` prints every key and the object tree for inspection when a path does not
resolve.
"""
import argparse
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
OBJECT_GRAPH_KEY = '_CHECKPOINTABLE_OBJECT_GRAPH'
DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 4: np.dtype('u1'), 5: np.dtype('<i2'), 6: np.dtype('i1'),
          9: np.dtype('<i8'), 10: np.dtype('?'), 17: np.dtype('<u2'), 19: np.dtype('<f2'), 22: np.dtype('<u4'), 23: np.dtype('<u8')}
DT_STRING = 7


class CheckpointError(RuntimeError):
    pass


# ------------------------------------------------------------------------------------------------ wire formats
def _varint(buf, pos):
    out, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7f) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def proto_fields(buf):
    """Protocol-buffer wire format: [(field number, value)], value an int (varint / fixed) or bytes (length-delimited)."""
    pos, out = 0, []
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(buf, pos)
        elif wire == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError(f'unsupported protobuf wire type {wire}')
        out.append((field, v))
    return out


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of table blocks and tensor data."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82f63b78 if c & 1 else 0)
            t.append(c)
        _CRC_TABLE = t
    c = crc ^ 0xffffffff
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


# ------------------------------------------------------------------------------------------------ string table
def _block(buf, offset, size, verify):
    body = buf[offset:offset + size]
    if len(body) != size or offset + size + 5 > len(buf):
        raise CheckpointError('table block runs past the end of the index file')
    kind = buf[offset + size]
    if kind != 0:
        raise CheckpointError(f'table block compression type {kind} is not supported (bundles are written uncompressed)')
    if verify:
        want = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if mask_crc(crc32c(buf[offset:offset + size + 1])) != want:
            raise CheckpointError('table block checksum mismatch')
    return body


def _block_entries(body):
    n_restarts = struct.unpack_from('<I', body, len(body) - 4)[0]
    end = len(body) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _varint(body, pos)
        unshared, pos = _varint(body, pos)
        vlen, pos = _varint(body, pos)
        key = key[:shared] + bytes(body[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(body[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise CheckpointError(f'{path} is not a tensor-bundle index (bad magic)')
    footer = buf[len(buf) - 48:]
    pos = 0
    _, pos = _varint(footer, pos)          # metaindex handle
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_block(buf, ioff, isize, verify)):
        off, p = _varint(handle, 0)
        size, _ = _varint(handle, p)
        out.extend(_block_entries(_block(buf, off, size, verify)))
    return out


# ------------------------------------------------------------------------------------------------ tensor bundle
class Bundle:
    """Read access to one checkpoint: `keys()`, `entry(key)` -> (dtype code, shape, shard, offset, size, crc), `tensor(key)`."""

    def __init__(self, prefix, verify=True):
        if not os.path.exists(prefix + '.index'):
            raise CheckpointError(f'{prefix}.index not found')
        self.prefix = prefix
        self.entries = {}
        self.num_shards = 1
        for key, value in read_table(prefix + '.index', verify):
            f = proto_fields(value)
            if key == b'':
                head = dict(f)
                self.num_shards = head.get(1, 1)
                if head.get(2, 0) != 0:
                    raise CheckpointError('big-endian bundles are not supported')
                continue
            d = {'dtype': 0, 'shape': (), 'shard': 0, 'offset': 0, 'size': 0, 'crc': None}
            for no, v in f:
                if no == 1:
                    d['dtype'] = v
                elif no == 2:
                    d['shape'] = tuple(dict(proto_fields(dim)).get(1, 0) for n2, dim in proto_fields(v) if n2 == 2)
                elif no == 3:
                    d['shard'] = v
                elif no == 4:
                    d['offset'] = v
                elif no == 5:
                    d['size'] = v
                elif no == 6:
                    d['crc'] = v
                elif no == 7:
                    raise CheckpointError(f'{key.decode()}: sliced (partitioned) tensors are not supported')
            self.entries[key.decode()] = d

    def keys(self):
        return list(self.entries)

    def entry(self, key):
        return self.entries[key]

    def _raw(self, e):
        path = f'{self.prefix}.data-{e["shard"]:05d}-of-{self.num_shards:05d}'
        with open(path, 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise CheckpointError(f'{path} is shorter than the index says')
        return raw

    def tensor(self, key, verify=False):
        e = self.entries[key]
        raw = self._raw(e)
        if e['dtype'] == DT_STRING:
            return self._strings(raw, e)
        if e['dtype'] not in DTYPES:
            raise CheckpointError(f'{key}: tensor dtype code {e["dtype"]} is not supported')
        if verify and e['crc'] is not None and mask_crc(crc32c(raw)) != e['crc']:
            raise CheckpointError(f'{key}: data checksum mismatch')
        a = np.frombuffer(raw, dtype=DTYPES[e['dtype']])
        if a.size != int(np.prod(e['shape'], dtype=np.int64)):
            raise CheckpointError(f'{key}: {a.size} elements stored, shape {e["shape"]}')
        return a.reshape(e['shape']).copy()

    @staticmethod
    def _strings(raw, e):
        """String tensors: one varint length per element, a 4-byte checksum of the lengths, then the bytes."""
        n = int(np.prod(e['shape'], dtype=np.int64))
        pos, lens = 0, []
        for _ in range(n):
            v, pos = _varint(raw, pos)
            lens.append(v)
        if len(raw) - pos == sum(lens) + 4:
            pos += 4
        elif len(raw) - pos != sum(lens):
            raise CheckpointError('string tensor layout not understood')
        out = []
        for v in lens:
            out.append(raw[pos:pos + v])
            pos += v
        return out[0] if e['shape'] == () else out


class ObjectGraph:
    """TrackableObjectGraph: nodes[i] = {'children': {local name: node id}, 'attributes': {name: checkpoint key}}; node 0 is the
    root (the tf.train.Checkpoint object)."""

    def __init__(self, serialized):
        self.nodes = []
        for no, node in proto_fields(serialized):
            if no != 1:
                continue
            children, attributes = {}, {}
            for f, v in proto_fields(node):
                if f == 1:
                    ref = dict(proto_fields(v))
                    children[ref.get(2, b'').decode()] = ref.get(1, 0)
                elif f == 2:
                    att = dict(proto_fields(v))
                    attributes[att.get(1, b'').decode()] = att.get(3, b'').decode()
            self.nodes.append({'children': children, 'attributes': attributes})
        if not self.nodes:
            raise CheckpointError('empty object graph')

    def resolve(self, path):
        """Node id at the end of `path` (attribute names from the root)."""
        node = 0
        for i, name in enumerate(path):
            kids = self.nodes[node]['children']
            if name not in kids:
                raise CheckpointError(f'object {"/".join(path[:i]) or "<root>"} has no child {name!r}; it has {sorted(kids)}')
            node = kids[name]
        return node

    def variable_key(self, path):
        att = self.nodes[self.resolve(path)]['attributes']
        if 'VARIABLE_VALUE' not in att:
            raise CheckpointError(f'{"/".join(path)} is not a variable (attributes {sorted(att)})')
        return att['VARIABLE_VALUE']

    def tree(self, node=0, prefix='', seen=None):
        """Lines 'path -> checkpoint key' for every variable reachable from `node` (first path found, depth first)."""
        seen = set() if seen is None else seen
        if node in seen:
            return []
        seen.add(node)
        out = [f'{prefix or "<root>"} -> {k}' for a, k in self.nodes[node]['attributes'].items() if a == 'VARIABLE_VALUE']
        for name, child in self.nodes[node]['children'].items():
            out += self.tree(child, f'{prefix}/{name}' if prefix else name, seen)
        return out


# ------------------------------------------------------------------------------------------------ attribute paths
def object_path(key, model_kind, cfg):
    """Attribute path (below the model object) of one weight-dictionary key, e.g. for the forward model
    'decoder.block3.conv.norm1.gamma' -> decoder / encoder_SACB / <3 - dense blocks> / conv / normalization / 1 / gamma."""
    parts = key.split('.')
    top = parts[0]
    out = [top]
    for tok in parts[1:]:
        m = re.fullmatch(r'(block|conv|norm)(\d+)', tok)
        if m and m.group(1) == 'block':
            i = int(m.group(2))
            if model_kind == 'autoregressive' and top == 'decoder':                  # CrossAttentionBlocks, layers.py:350-358
                dense, names = int(cfg['decoder_dense_blocks']), ('CADB', 'CACB')
            else:                                                                    # SelfAttentionBlocks, layers.py:232-240
                dense = int(cfg['decoder_dense_blocks'] if top == 'decoder' else cfg['encoder_dense_blocks'])
                names = ('encoder_SADB', 'encoder_SACB')
            out += [names[0], str(i)] if i < dense else [names[1], str(i - dense)]
        elif m and m.group(1) == 'conv':
            out += ['convolutions', m.group(2)]                                      # layers.py:19, 538
        elif m:
            out += ['normalization', m.group(2)]                                     # layers.py:29-31, 542
        elif tok == 'gru':
            out += ['rnn_net', 'cell']                                               # layers.py:543
        elif tok == 'rnn_proj':
            out.append('rnn_proj_net')                                               # layers.py:545
        else:
            out.append(tok)
    return out


def template(model_kind, cfg):
    from . import weights
    init = weights.init_autoregressive_transformer if model_kind == 'autoregressive' else weights.init_forward_transformer
    return init(cfg, seed=0)


def load_tf_checkpoint(prefix, model_kind, cfg, root='net', verify=False):
    """Weight dictionary (this package's key names and Keras array layouts) from the checkpoint `<prefix>.index/.data-*`.
    cfg: the model configuration (decides block counts and the expected shapes).  Returns (weights, step or None)."""
    bundle = Bundle(prefix)
    if OBJECT_GRAPH_KEY not in bundle.entries:
        raise CheckpointError(f'{prefix}: no {OBJECT_GRAPH_KEY} entry: not an object-based (TF2) checkpoint')
    graph = ObjectGraph(bundle.tensor(OBJECT_GRAPH_KEY))
    out = {}
    for key, like in template(model_kind, cfg).items():
        ck = graph.variable_key([root] + object_path(key, model_kind, cfg))
        if ck not in bundle.entries:
            raise CheckpointError(f'{key}: object graph names {ck}, which the index does not hold')
        a = bundle.tensor(ck, verify)
        if tuple(a.shape) != tuple(np.shape(like)):
            raise CheckpointError(f'{key}: checkpoint shape {tuple(a.shape)}, this configuration expects {tuple(np.shape(like))}')
        out[key] = a.astype(np.asarray(like).dtype)
    step = None
    if 'step' in graph.nodes[0]['children']:
        try:
            step = int(bundle.tensor(graph.variable_key(['step'])))
        except CheckpointError:
            step = None
    return out, step


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('prefix', help='checkpoint prefix, e.g. .../autoregressive_weights/ckpt-90')
    ap.add_argument('--list', action='store_true', help='print the index keys and the object tree, convert nothing')
    ap.add_argument('--config', help='configuration directory (autoregressive_config.yaml / forward_config.yaml / data_config.yaml)')
    ap.add_argument('--model', choices=['forward', 'autoregressive'], default='forward')
    ap.add_argument('--out', help='.npz to write (the format ConfigManager.load_model reads)')
    ap.add_argument('--verify', action='store_true', help='check the CRC-32C of every tensor (slow in pure Python)')
    args = ap.parse_args()
    if args.list:
        b = Bundle(args.prefix)
        for k in b.keys():
            e = b.entry(k)
            print(f'{k}  dtype {e["dtype"]} shape {e["shape"]}')
        if OBJECT_GRAPH_KEY in b.entries:
            print('\n'.join(ObjectGraph(b.tensor(OBJECT_GRAPH_KEY)).tree()))
        return
    if not args.config or not args.out:
        ap.error('--config and --out are required unless --list is given')
    from .config_manager import ConfigManager
    cm = ConfigManager(args.config, args.model)
    w, step = load_tf_checkpoint(args.prefix, args.model, cm.model_config_dict(), verify=args.verify)
    cm.save_weights(args.out, w, step or 0)
    print(f'{len(w)} tensors, step {step} -> {args.out}')


if __name__ == '__main__':
    main()
