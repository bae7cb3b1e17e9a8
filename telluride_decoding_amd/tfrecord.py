"""Dependency-free TFRecord / tf.train.Example reader and writer.

The step before the hot path: the reference stores every recording as a TFRecord file of
tf.train.Example protos, ONE EXAMPLE PER FRAME, each feature a FloatList (writer
ingest.BrainTrial.write_data_as_tfrecords, ingest.py:612-651; reader
brain_data.TFExampleData, brain_data.py:733-839, discover_feature_shapes :887-927,
count_tfrecords :930-956).  TensorFlow is not a dependency of this package, so the two wire
formats are parsed here directly:

  TFRecord framing   uint64 length | uint32 masked-crc32c(length) | data | uint32 masked-crc32c(data)
  tf.train.Example   message Example  { Features features = 1; }
                     message Features { map<string, Feature> feature = 1; }
                     message Feature  { oneof kind { BytesList bytes_list = 1;
                                                     FloatList float_list = 2;
                                                     Int64List int64_list = 3; } }
                     FloatList / Int64List: repeated value = 1 (packed or not)

read_file() returns {feature name: float32 [frames, width]} -- the row-major time x channel
arrays the kernels take; dataset_from_files() applies the reference's field selection
(parse_and_select_from_tfrecord, brain_data.py:777-839: concatenated in1/in2 fields, the 'ones'
pseudo output, the placeholder input_2 / attended_speaker) and returns a brain_data.Dataset.
"""
import os
import struct

import numpy as np

_MASK_DELTA = 0xa282ead8


def _make_crc_table():
  poly = 0x82f63b78            # CRC-32C (Castagnoli), reflected
  table = []
  for i in range(256):
    c = i
    for _ in range(8):
      c = (c >> 1) ^ poly if c & 1 else c >> 1
    table.append(c)
  return table


_CRC_TABLE = _make_crc_table()


def crc32c(data):
  crc = 0xffffffff
  tab = _CRC_TABLE
  for b in data:
    crc = tab[(crc ^ b) & 0xff] ^ (crc >> 8)
  return crc ^ 0xffffffff


def masked_crc32c(data):
  crc = crc32c(data)
  return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + _MASK_DELTA) & 0xffffffff


# ---------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
  result, shift = 0, 0
  while True:
    b = buf[pos]
    pos += 1
    result |= (b & 0x7f) << shift
    if not b & 0x80:
      return result, pos
    shift += 7


def _fields(buf):
  """Yields (field number, wire type, value) of one message; length-delimited values are
  memoryview slices."""
  pos, end = 0, len(buf)
  while pos < end:
    key, pos = _varint(buf, pos)
    num, wt = key >> 3, key & 7
    if wt == 0:
      val, pos = _varint(buf, pos)
    elif wt == 1:
      val, pos = buf[pos:pos + 8], pos + 8
    elif wt == 2:
      n, pos = _varint(buf, pos)
      val, pos = buf[pos:pos + n], pos + n
    elif wt == 5:
      val, pos = buf[pos:pos + 4], pos + 4
    else:
      raise ValueError('unsupported protobuf wire type %d' % wt)
    if pos > end:
      raise ValueError('truncated protobuf message')
    yield num, wt, val


def _parse_feature(buf):
  """Feature -> np.ndarray (float32 / int64) or list of bytes."""
  for num, wt, val in _fields(buf):
    if wt != 2:
      continue
    if num == 2:      # FloatList
      out = []
      for n2, w2, v2 in _fields(val):
        if n2 == 1 and w2 == 2:                      # packed
          out.append(np.frombuffer(v2, dtype='<f4'))
        elif n2 == 1 and w2 == 5:                    # one value
          out.append(np.frombuffer(v2, dtype='<f4'))
      return np.concatenate(out) if out else np.zeros((0,), np.float32)
    if num == 3:      # Int64List
      out = []
      for n2, w2, v2 in _fields(val):
        if n2 == 1 and w2 == 2:
          p, vals = 0, []
          while p < len(v2):
            x, p = _varint(v2, p)
            vals.append(x - (1 << 64) if x >= (1 << 63) else x)
          out.extend(vals)
        elif n2 == 1 and w2 == 0:
          out.append(v2 - (1 << 64) if v2 >= (1 << 63) else v2)
      return np.asarray(out, np.int64)
    if num == 1:      # BytesList
      return [bytes(v2) for n2, w2, v2 in _fields(val) if n2 == 1 and w2 == 2]
  return np.zeros((0,), np.float32)


def parse_example(record):
  """One serialized tf.train.Example -> {name: values}."""
  out = {}
  buf = memoryview(record)
  for num, wt, features in _fields(buf):
    if num != 1 or wt != 2:
      continue
    for n2, w2, entry in _fields(features):          # map<string, Feature> entries
      if n2 != 1 or w2 != 2:
        continue
      key, feat = None, None
      for n3, w3, v3 in _fields(entry):
        if n3 == 1 and w3 == 2:
          key = bytes(v3).decode('utf-8')
        elif n3 == 2 and w3 == 2:
          feat = v3
      if key is not None:
        out[key] = _parse_feature(feat) if feat is not None else np.zeros((0,), np.float32)
  return out


# ---------------------------------------------------------------- TFRecord framing
def iter_records(filename, verify=False):
  """Yields the raw records of a TFRecord file.  verify=True checks both CRCs."""
  with open(filename, 'rb') as f:
    data = f.read()
  pos, end = 0, len(data)
  view = memoryview(data)
  while pos < end:
    if pos + 12 > end:
      raise ValueError('%s: truncated record header at byte %d' % (filename, pos))
    (length,) = struct.unpack_from('<Q', data, pos)
    if verify and struct.unpack_from('<I', data, pos + 8)[0] != masked_crc32c(view[pos:pos + 8]):
      raise ValueError('%s: corrupt length CRC at byte %d' % (filename, pos))
    start = pos + 12
    if start + length + 4 > end:
      raise ValueError('%s: truncated record at byte %d' % (filename, pos))
    rec = view[start:start + length]
    if verify and struct.unpack_from('<I', data, start + length)[0] != masked_crc32c(rec):
      raise ValueError('%s: corrupt data CRC at byte %d' % (filename, pos))
    yield rec
    pos = start + length + 4


def count_tfrecords(tfrecord_file_name):
  """(valid records, error found) like brain_data.count_tfrecords (brain_data.py:930-956)."""
  if not isinstance(tfrecord_file_name, str):
    raise TypeError('tfrecord_file_name must be a string.')
  count = 0
  try:
    for rec in iter_records(tfrecord_file_name):
      parse_example(rec)
      count += 1
  except Exception:   # pylint: disable=broad-except
    return count, True
  return count, False


def discover_feature_shapes(tfrecord_file_name):
  """{feature name: (width, dtype)} of the first record (brain_data.py:887-927)."""
  if not isinstance(tfrecord_file_name, str):
    raise TypeError('discover_feature_shapes: input must be a string filename.')
  for rec in iter_records(tfrecord_file_name):
    ex = parse_example(rec)
    return {k: (len(v), 'bytes' if isinstance(v, list) else str(v.dtype)) for k, v in ex.items()}
  return {}


def _float_layout(record):
  """[(feature name, byte offset of its packed float payload in the record, float count)] when
  every feature of the Example is ONE packed FloatList (what the reference's ingest writes per
  frame), else None."""
  def fields_at(buf, base):
    pos, end = 0, len(buf)
    while pos < end:
      key, pos = _varint(buf, pos)
      num, wt = key >> 3, key & 7
      if wt != 2:
        raise ValueError('not a length-delimited field')
      n, pos = _varint(buf, pos)
      if pos + n > end:
        raise ValueError('truncated')
      yield num, buf[pos:pos + n], base + pos
      pos += n
  out = []
  try:
    buf = memoryview(record)
    for num, features, f0 in fields_at(buf, 0):
      if num != 1:
        return None
      for n2, entry, e0 in fields_at(features, f0):
        if n2 != 1:
          return None
        key, payload = None, None
        for n3, v3, v0 in fields_at(entry, e0):
          if n3 == 1:
            key = bytes(v3).decode('utf-8')
          elif n3 == 2:
            inner = list(fields_at(v3, v0))
            if len(inner) != 1 or inner[0][0] != 2:          # exactly one FloatList
              return None
            lists = list(fields_at(inner[0][1], inner[0][2]))
            if len(lists) != 1 or lists[0][0] != 1 or len(lists[0][1]) % 4:
              return None
            payload = (lists[0][2], len(lists[0][1]) // 4)
          else:
            return None
        if key is None or payload is None:
          return None
        out.append((key, payload[0], payload[1]))
  except (ValueError, IndexError, UnicodeDecodeError):
    return None
  return out or None


def _read_file_regular(filename, fields):
  """read_file for the regular case -- every record the same length and the same bytes outside
  its float payloads (one Example per frame, fixed feature widths) -- as array slicing of the whole
  file: no per-record Python (31 us a record through the generic parser: half a minute per 1e6
  frames).  None when the file is not of that shape."""
  size = os.path.getsize(filename)
  if size < 16:
    return None
  # the file is mapped, not read: the only copies are the feature blocks that are asked for (peak
  # host memory ~1x the payload instead of ~3x the file)
  data = np.memmap(filename, dtype=np.uint8, mode='r')
  (length,) = struct.unpack_from('<Q', bytes(data[:8]), 0)
  stride = length + 16
  if length == 0 or size % stride:
    return None
  layout = _float_layout(memoryview(bytes(data[12:12 + length])))
  if layout is None:
    return None
  arr = data.reshape(-1, stride)
  skeleton = np.ones(stride, bool)
  skeleton[8:12] = False                       # (CRC of the length: equal anyway)
  skeleton[12 + length:] = False               # CRC of the data
  for _, start, count in layout:
    skeleton[12 + start:12 + start + 4 * count] = False
  # every record's bytes outside its payloads against the first record's, in chunks of rows with an
  # early exit (a boolean copy of the whole file's skeleton was 1x the file again)
  cols = np.flatnonzero(skeleton)
  first = np.asarray(arr[0, cols])
  chunk = max(1, (8 << 20) // max(1, len(cols)))
  for r0 in range(0, arr.shape[0], chunk):
    if not np.array_equal(arr[r0:r0 + chunk][:, cols], np.broadcast_to(first, (min(chunk, arr.shape[0] - r0), len(cols)))):
      return None
  out = {}
  for key, start, count in layout:
    if fields is not None and key not in fields:
      continue
    if key in out:
      return None
    block = np.ascontiguousarray(arr[:, 12 + start:12 + start + 4 * count])
    out[key] = block.view('<f4').astype(np.float32, copy=False)
  return out


def read_file(filename, fields=None, verify=False):
  """{feature: float32 [frames, width]} of the float / int features (one Example per frame)."""
  if not verify:
    out = _read_file_regular(filename, fields)
    if out is not None:
      if fields is not None:
        missing = set(fields) - set(out)
        if missing:
          raise ValueError('Could not find all desired features (%s) in data (%s)' %
                           (sorted(fields), sorted(out)))
      return out
  cols = {}
  for rec in iter_records(filename, verify=verify):
    ex = parse_example(rec)
    for k, v in ex.items():
      if isinstance(v, list) or (fields is not None and k not in fields):
        continue
      cols.setdefault(k, []).append(v)
  out = {}
  for k, rows in cols.items():
    widths = {len(r) for r in rows}
    if len(widths) != 1:
      raise ValueError('%s: feature %s changes width (%s)' % (filename, k, sorted(widths)))
    out[k] = np.stack(rows).astype(np.float32)
  if fields is not None:
    missing = set(fields) - set(out)
    if missing:
      raise ValueError('Could not find all desired features (%s) in data (%s)' %
                       (sorted(fields), sorted(out)))
  return out


# ---------------------------------------------------------------- writer
def _enc_varint(x):
  out = bytearray()
  while True:
    b = x & 0x7f
    x >>= 7
    if x:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def _ld(num, payload):
  return _enc_varint((num << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
  """{name: 1-D float array} -> serialized tf.train.Example with packed FloatLists."""
  entries = b''
  for name in sorted(features):
    vals = np.asarray(features[name], '<f4').reshape(-1)
    float_list = _ld(1, vals.tobytes())
    feature = _ld(2, float_list)
    entries += _ld(1, _ld(1, name.encode('utf-8')) + _ld(2, feature))
  return _ld(1, entries)


def write_file(filename, data):
  """data: {name: [frames, width]} -> TFRecord file, one Example per frame
  (ingest.BrainTrial.write_data_as_tfrecords, ingest.py:612-651)."""
  n = {np.asarray(v).shape[0] for v in data.values()}
  if len(n) != 1:
    raise ValueError('all features need the same number of frames, not %s' % sorted(n))
  arrays = {k: np.asarray(v, np.float32).reshape(np.asarray(v).shape[0], -1) for k, v in data.items()}
  with open(filename, 'wb') as f:
    for i in range(n.pop()):
      rec = serialize_example({k: a[i] for k, a in arrays.items()})
      head = struct.pack('<Q', len(rec))
      f.write(head + struct.pack('<I', masked_crc32c(head)) + rec +
              struct.pack('<I', masked_crc32c(rec)))


# ---------------------------------------------------------------- field selection
def select_streams(features, in1_fields, out_field, in2_fields=None, attended_field=None):
  """(input_1, input_2, output, attended_speaker) of one recording, as
  brain_data.TFExampleData.parse_and_select_from_tfrecord builds them per frame
  (brain_data.py:777-839): listed fields concatenated; out_field 'ones' = a column of ones;
  no input_2 fields -> the first column of input_1; no attended field -> zeros."""
  in1_fields = [in1_fields] if isinstance(in1_fields, str) else list(in1_fields)
  in2_fields = [] if not in2_fields else ([in2_fields] if isinstance(in2_fields, str)
                                          else list(in2_fields))
  missing = set(in1_fields) - set(features)
  if missing:
    raise ValueError('Could not find all desired features (%s) in data (%s)' %
                     (in1_fields, sorted(features)))
  x = np.concatenate([features[k] for k in in1_fields], axis=1)
  if out_field == 'ones':
    y = np.ones((x.shape[0], 1), np.float32)
  else:
    y = features[out_field]
  if in2_fields:
    for k in in2_fields:
      if k not in features:
        raise ValueError('Could not find %s in parsed_features[%s]' % (k, sorted(features)))
    x2 = np.concatenate([features[k] for k in in2_fields], axis=1)
  else:
    x2 = x[:, 0:1]
  att = features[attended_field] if attended_field else np.zeros((x.shape[0], 1), np.float32)
  return x, x2, y, att


def dataset_from_files(filenames, in1_fields, out_field, in2_fields=None, attended_field=None,
                       batch_size=512, pre_context=0, post_context=0, in2_pre_context=0,
                       in2_post_context=0, input_offset=0):
  """TFRecord files -> brain_data.Dataset (one file = one recording; context never crosses
  files, brain_data.py:722-724).  Files whose name contains '-bad-' are skipped (:677)."""
  from telluride_decoding_amd import brain_data
  wanted = set([in1_fields] if isinstance(in1_fields, str) else in1_fields)
  wanted |= set([] if not in2_fields else ([in2_fields] if isinstance(in2_fields, str) else in2_fields))
  if out_field != 'ones':
    wanted.add(out_field)
  if attended_field:
    wanted.add(attended_field)
  files = []
  for name in filenames:
    if '-bad-' in name:
      continue
    files.append(select_streams(read_file(name, fields=wanted), in1_fields, out_field, in2_fields,
                                attended_field))
  return brain_data.Dataset(files, batch_size, pre_context=pre_context, post_context=post_context,
                            in2_pre_context=in2_pre_context, in2_post_context=in2_post_context,
                            input_offset=input_offset)
