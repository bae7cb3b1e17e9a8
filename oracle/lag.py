"""A1 -- temporal-context ("lag") matrix.  Test infrastructure (see oracle/__init__).

Restates brain_data.BrainData.add_temporal_context.window_one_stream_new
(telluride_decoding/brain_data.py:425-457) and window_data (:459-483).
"""
import numpy as np


def lag_matrix(x, pre, post):
  """Row t, column l*C + c holds x~[t + l - pre, c]; x~ is zero outside [0, N).

  brain_data.py:448-452 zero-pads `pre` rows in front and `post` rows behind,
  :453 frames with length pre+1+post and step 1, :454 flattens each frame
  (lag-major, channel-minor).  N rows in, N rows out.
  """
  x = np.asarray(x)
  n, c = x.shape
  width = pre + 1 + post
  padded = np.concatenate(
      [np.zeros((pre, c), x.dtype), x, np.zeros((post, c), x.dtype)], axis=0)
  out = np.empty((n, width * c), x.dtype)
  for l in range(width):
    out[:, l * c:(l + 1) * c] = padded[l:l + n]
  return out


def window_streams(x, x2, y, a, pre=0, post=0, pre2=0, post2=0, input_offset=0):
  """Four aligned streams of one file after offsets and context.

  brain_data.py:466-475: a positive offset drops leading rows of x, a negative
  one drops leading rows of x2 and y (not of the attention stream); :477-483
  zip truncates all streams to the shortest.
  """
  x, x2, y, a = (np.asarray(v) for v in (x, x2, y, a))
  if input_offset > 0:
    x = x[input_offset:]
  elif input_offset < 0:
    x2 = x2[-input_offset:]
    y = y[-input_offset:]
  xl = lag_matrix(x, pre, post)
  x2l = lag_matrix(x2, pre2, post2)
  n = min(xl.shape[0], x2l.shape[0], y.shape[0], a.shape[0])
  return xl[:n], x2l[:n], y[:n], a[:n]


def minibatches(files, batch_size, pre=0, post=0, pre2=0, post2=0,
                input_offset=0):
  """Concatenate the per-file streams and cut them into full minibatches.

  Context is added per file (brain_data.py:722-724 / :492-500) and the stream
  is then batched with drop_remainder=True (brain_data.py:369-370), so only
  the tail of the *whole* stream is lost.  `files` is a list of
  (x, x2, y, attention) tuples.  Yields (dict, y) like the reference dataset
  (brain_data.py:386-390).
  """
  parts = [window_streams(*f, pre=pre, post=post, pre2=pre2, post2=post2,
                          input_offset=input_offset) for f in files]
  xs, x2s, ys, as_ = (np.concatenate([p[i] for p in parts], axis=0)
                      for i in range(4))
  n_batches = xs.shape[0] // batch_size
  for b in range(n_batches):
    s = slice(b * batch_size, (b + 1) * batch_size)
    yield ({'input_1': xs[s], 'input_2': x2s[s], 'attended_speaker': as_[s]},
           ys[s])
