"""Seeded synthetic EEG / speech-envelope generator for tests and bench.

Builder's own generator (NumPy, host side, never timed).  It mirrors the
structure of the reference's simulated-EEG fixture
(test/brain_model_test.py:575-726, test/decoding_test.py:135-216): speech
envelopes are low-pass noise (noise at fs/10 up-sampled x10), each channel's
response is a random 250 ms impulse response shaped by 30*t*exp(-30*t), and
EEG = attended * h_att + 0.1 * (unattended * h_unatt) + 0.3 * N(0, 1).
"""
import numpy as np

FRAME_RATE = 100  # Hz, the reference's default frame_rate (decoding.py flags)


def impulse_responses(rng, num_channels, fs=FRAME_RATE, length_s=0.25,
                      unattended_gain=0.1):
  t = np.arange(int(round(length_s * fs))) / float(fs)
  shape = (30.0 * t * np.exp(-30.0 * t)).reshape(-1, 1)
  h_att = rng.standard_normal((t.shape[0], num_channels)) * shape
  h_unatt = rng.standard_normal((t.shape[0], num_channels)) * shape * unattended_gain
  return h_att, h_unatt


def envelopes(rng, num_frames, num_streams=2, upsample=10):
  """Low-pass random envelopes: noise at fs/upsample, linearly interpolated."""
  n_low = int(np.ceil(num_frames / float(upsample))) + 1
  low = rng.standard_normal((n_low, num_streams))
  t = np.arange(num_frames) / float(upsample)
  i0 = np.floor(t).astype(np.int64)
  frac = (t - i0).reshape(-1, 1)
  return ((1.0 - frac) * low[i0] + frac * low[i0 + 1]).astype(np.float32)


def _causal_conv(sig, h):
  """sig [N], h [T, C] -> [N, C] (first N samples of the full convolution)."""
  n = sig.shape[0]
  nfft = 1
  while nfft < n + h.shape[0]:
    nfft *= 2
  spec = np.fft.rfft(sig.astype(np.float64), nfft).reshape(-1, 1)
  out = np.fft.irfft(spec * np.fft.rfft(h, nfft, axis=0), nfft, axis=0)
  return out[:n]


def trial(rng, num_frames, num_channels, h_att, h_unatt, attention=None,
          noise_level=0.3):
  """One trial.  Returns (eeg [N,C] f32, env [N,2] f32, attended [N,1] f32).

  `attention[t]` is 0 when speaker 1 is attended and 1 for speaker 2 (the
  reference's `attended_speaker` convention, infer.py:402-407).
  """
  env = envelopes(rng, num_frames, 2)
  if attention is None:
    attention = np.zeros((num_frames,), np.float32)
  attention = np.asarray(attention, np.float32).reshape(-1)
  att_audio = np.where(attention > 0.5, env[:, 1], env[:, 0])
  unatt_audio = np.where(attention > 0.5, env[:, 0], env[:, 1])
  eeg = (_causal_conv(att_audio, h_att) + _causal_conv(unatt_audio, h_unatt)
         + noise_level * rng.standard_normal((num_frames, num_channels)))
  return (eeg.astype(np.float32), env,
          attention.reshape(-1, 1).astype(np.float32))


def make_trials(seed, num_trials, num_frames, num_channels, switch_half=False):
  """A list of trials sharing one pair of impulse responses.

  With `switch_half`, odd-numbered trials switch attention at mid-trial
  (SURVEY.md 8d, config C4).
  """
  rng = np.random.default_rng(seed)
  h_att, h_unatt = impulse_responses(rng, num_channels)
  out = []
  for i in range(num_trials):
    att = np.zeros((num_frames,), np.float32)
    if switch_half and (i % 2 == 1):
      att[num_frames // 2:] = 1.0
    out.append(trial(rng, num_frames, num_channels, h_att, h_unatt, att))
  return out
