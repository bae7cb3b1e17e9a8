"""F3: the dependency-free TFRecord / tf.train.Example reader."""
import glob
import os

import numpy as np
import pytest

from telluride_decoding_amd import tfrecord

REF_MEG = '/root/reference/test_data/meg'


def test_crc32c_known_answers():
  # RFC 3720 B.4 test vectors
  assert tfrecord.crc32c(b'\x00' * 32) == 0x8a9136aa
  assert tfrecord.crc32c(b'\xff' * 32) == 0x62a8ab43
  assert tfrecord.crc32c(bytes(range(32))) == 0x46dd794e
  assert tfrecord.crc32c(b'123456789') == 0xe3069283


def test_round_trip_and_corruption(tmp_path):
  rng = np.random.default_rng(0)
  data = {'eeg': rng.standard_normal((37, 5)).astype(np.float32),
          'envelope': rng.standard_normal((37, 1)).astype(np.float32),
          'attended': (rng.random((37, 1)) > 0.5).astype(np.float32)}
  name = str(tmp_path / 'trial.tfrecords')
  tfrecord.write_file(name, data)
  assert tfrecord.count_tfrecords(name) == (37, False)
  shapes = tfrecord.discover_feature_shapes(name)
  assert shapes == {'eeg': (5, 'float32'), 'envelope': (1, 'float32'), 'attended': (1, 'float32')}
  back = tfrecord.read_file(name, verify=True)
  for k in data:
    np.testing.assert_array_equal(back[k], data[k])
  with pytest.raises(ValueError, match='Could not find all desired features'):
    tfrecord.read_file(name, fields=['meg'])
  with pytest.raises(TypeError):
    tfrecord.count_tfrecords(3)
  # a flipped payload byte is caught by the data CRC, a truncated file by the framing
  raw = bytearray(open(name, 'rb').read())
  raw[40] ^= 0xff
  bad = str(tmp_path / 'bad.tfrecords')
  open(bad, 'wb').write(bytes(raw))
  with pytest.raises(ValueError, match='corrupt'):
    list(tfrecord.iter_records(bad, verify=True))
  open(bad, 'wb').write(open(name, 'rb').read()[:-7])
  assert tfrecord.count_tfrecords(bad) == (36, True)


def test_regular_files_are_read_without_a_loop_over_records(tmp_path):
  """One Example per frame with fixed widths (what the reference's ingest writes) is read by
  slicing the file as an array; anything else -- a record of another width, a corrupt skeleton
  byte -- falls back to the record-by-record parser, and both agree."""
  rng = np.random.default_rng(2)
  data = {'eeg': rng.standard_normal((500, 7)).astype(np.float32),
          'envelope': rng.standard_normal((500, 1)).astype(np.float32)}
  name = str(tmp_path / 'regular.tfrecords')
  tfrecord.write_file(name, data)
  fast = tfrecord._read_file_regular(name, None)
  assert fast is not None
  slow = tfrecord.read_file(name, verify=True)            # (verification takes the generic path)
  for k in data:
    np.testing.assert_array_equal(fast[k], data[k])
    np.testing.assert_array_equal(slow[k], data[k])
    assert fast[k].dtype == np.float32 and fast[k].flags['C_CONTIGUOUS']
  np.testing.assert_array_equal(tfrecord.read_file(name, fields=['eeg'])['eeg'], data['eeg'])
  with pytest.raises(ValueError, match='Could not find all desired features'):
    tfrecord.read_file(name, fields=['meg'])
  here = os.path.join(os.path.dirname(__file__), 'golden', 'meg_subj01_400.tfrecords')
  a, b = tfrecord.read_file(here), tfrecord.read_file(here, verify=True)
  assert tfrecord._read_file_regular(here, None) is not None and set(a) == set(b)
  for k in a:
    np.testing.assert_array_equal(a[k], b[k])
  # two files with different widths appended: not regular, the generic parser reports the change
  other = str(tmp_path / 'other.tfrecords')
  tfrecord.write_file(other, {'eeg': data['eeg'][:3, :5], 'envelope': data['envelope'][:3]})
  mixed = str(tmp_path / 'mixed.tfrecords')
  open(mixed, 'wb').write(open(name, 'rb').read() + open(other, 'rb').read())
  assert tfrecord._read_file_regular(mixed, None) is None
  with pytest.raises(ValueError, match='changes width'):
    tfrecord.read_file(mixed)
  # a feature renamed in ONE record (same length): the skeleton check sends the file to the parser
  raw = bytearray(open(name, 'rb').read())
  at = raw.find(b'envelope', len(raw) // 2)
  raw[at] = ord('E')
  odd = str(tmp_path / 'odd.tfrecords')
  open(odd, 'wb').write(bytes(raw))
  assert tfrecord._read_file_regular(odd, None) is None


def test_field_selection_matches_reference_semantics(tmp_path):
  rng = np.random.default_rng(1)
  names = []
  for i, n in enumerate((120, 80)):
    data = {'eeg': rng.standard_normal((n, 4)).astype(np.float32),
            'extra': rng.standard_normal((n, 2)).astype(np.float32),
            'envelope': rng.standard_normal((n, 1)).astype(np.float32)}
    names.append(str(tmp_path / ('s%d.tfrecords' % i)))
    tfrecord.write_file(names[-1], data)
  names.append(str(tmp_path / 's9-bad-.tfrecords'))       # skipped like brain_data.py:677
  ds = tfrecord.dataset_from_files(names, ['eeg', 'extra'], 'envelope', batch_size=50,
                                   pre_context=1, post_context=2)
  assert len(ds.files) == 2 and ds.c1 == 6 and ds.d == 1
  x, x2, y, att = ds.files[0]
  np.testing.assert_array_equal(x2, x[:, 0:1])             # placeholder input_2
  assert not att.any()                                     # placeholder attended_speaker
  batches = list(ds)
  assert len(batches) == (120 + 80) // 50 and batches[0][0]['input_1'].shape == (50, 6 * 4)
  ones = tfrecord.dataset_from_files(names[:1], 'eeg', 'ones', in2_fields='envelope')
  assert ones.files[0][2].min() == ones.files[0][2].max() == 1.0 and ones.c2 == 1
  with pytest.raises(ValueError, match='Could not find'):
    tfrecord.dataset_from_files(names[:1], 'eeg', 'envelope', in2_fields=['nope'])


@pytest.mark.skipif(not os.path.isdir(REF_MEG), reason='reference test data not present')
def test_reads_the_reference_meg_files():
  """The reference's own fixtures and expectations (test/brain_data_test.py:501-531):
  1001 records per file; meg 148, mel_spectrogram 64, phonemes 38, phonetic_features 19,
  envelope 1 wide."""
  files = sorted(glob.glob(os.path.join(REF_MEG, '*.tfrecords')))
  assert len(files) == 3
  assert tfrecord.count_tfrecords(files[0]) == (1001, False)
  shapes = tfrecord.discover_feature_shapes(files[0])
  for k, w in {'phonetic_features': 19, 'mel_spectrogram': 64, 'meg': 148, 'phonemes': 38,
               'envelope': 1}.items():
    assert shapes[k][0] == w
  feats = tfrecord.read_file(files[0], fields=['meg', 'envelope'], verify=True)
  assert feats['meg'].shape == (1001, 148) and feats['envelope'].shape == (1001, 1)
  assert np.isfinite(feats['meg']).all() and feats['meg'].std() > 0
