"""A8/A9 -- attended-speaker decision.  Test infrastructure.

Restates attention_decoder.AttentionDecoder.attention
(telluride_decoding/attention_decoder.py:128-134), StepAttentionDecoder
(:141-173), StateSpaceAttentionDecoder (:176-451), create_attention_decoder
(:455-485) and the accuracy rule of infer.run_reduction_test
(telluride_decoding/infer.py:399-407).
"""
import numpy as np


def wta(r1, r2):
  """Strict '>' on the means; ties go to speaker 2 (attention_decoder.py:134)."""
  return bool(np.mean(r1) > np.mean(r2))


def wta_sequence(r1, r2):
  return np.array([wta(a, b) for a, b in zip(r1, r2)], dtype=bool)


def step_sequence(r1, r2, state=0.5):
  """attention_decoder.py:169-173: +-0.1 per call, clipped to [0.1, 0.9],
  decision is state > 0.5 (fp64; 0.9-4*0.1 = 0.5000000000000001 counts)."""
  out = []
  for a, b in zip(r1, r2):
    if np.mean(a) > np.mean(b):
      state = min(0.9, state + 0.1)
    else:
      state = max(0.1, state - 0.1)
    out.append(state > 0.5)
  return np.array(out, dtype=bool), state


def tune_log_normal_priors(r1, r2, offset=0.0):
  """attention_decoder.py:291-320.  Returns (rho_d, mu_d) (mu_0 = mu_d)."""
  a1 = np.absolute(np.asarray(r1, np.float64) + offset)
  a2 = np.absolute(np.asarray(r2, np.float64) + offset)
  n = a1.shape[0]
  u_a = np.sum(a1) / n
  v_a = np.sum((a1 - u_a) ** 2) / n
  rho_a = 1 / np.log(v_a / u_a ** 2 + 1)
  mu_a = np.log(u_a) - 0.5 / rho_a
  u_u = np.sum(a2) / n
  v_u = np.sum((a2 - u_u) ** 2) / n
  rho_u = 1 / np.log(v_u / u_u ** 2 + 1)
  mu_u = np.log(u_u) - 0.5 / rho_u
  return [rho_a, rho_u], [mu_a, mu_u]


class StateSpace(object):
  """Fixed-lag state-space decoder; one stream (attention_decoder.py:183-271)."""

  def __init__(self, outer_iter=20, inner_iter=1, newton_iter=10,
               forward_lag=0, backward_lag=13, offset=0.0):
    self.offset = offset
    self.outer_iter, self.inner_iter, self.newton_iter = (
        outer_iter, inner_iter, newton_iter)
    self.k_f, self.k_b = forward_lag, backward_lag
    self.k_w = self.k_f + self.k_b + 1                    # :217
    self.c0 = 1.96                                        # :222
    mean_p, var_p = 0.2, 5                                # :225-228
    self.a_0 = 2 + mean_p ** 2 / var_p
    self.b_0 = mean_p * (self.a_0 - 1)
    self.calls = 0
    self.r1, self.r2 = [], []
    self.z_smoothed = [0.0] * self.k_w                    # :244-248
    self.eta_smoothed = [0.3] * self.k_w
    self.z_dyn = [0.0] * self.k_w
    self.eta_dyn = [0.0] * self.k_w
    self.lam = 1.0                                        # :251
    n = self.k_w + 1
    self.z_kk = np.zeros(n)                               # :254-262
    self.s_kk = np.zeros(n)
    self.z_pred = np.zeros(n)
    self.s_pred = np.zeros(n)
    self.z_cap = np.zeros(n)
    self.s_cap = np.zeros(n)
    self.sm = np.zeros(self.k_w)
    self.alpha_0 = [6.4113e+02, 4.0434e+03]               # :266-271
    self.beta_0 = [3.7581e+02, 6.2791e+03]
    self.mu_0 = [-0.3994, -1.5103]
    self.rho_d = [1.7060, 0.64395]
    self.mu_d = [-0.3994, -1.5103]

  def tune(self, r1, r2):
    self.rho_d, self.mu_d = tune_log_normal_priors(r1, r2, self.offset)
    self.mu_0 = list(self.mu_d)                           # :320
    self.alpha_0 = [6.4113e+02, 4.0434e+03]
    self.beta_0 = [3.7581e+02, 6.2791e+03]

  def attention(self, r1, r2):
    self.calls += 1                                       # :350-352
    self.r1.append(np.abs(r1 + self.offset))
    self.r2.append(np.abs(r2 + self.offset))
    if self.calls < self.k_w:
      return (0.5, 0.5, 0.5)                              # :451
    kw = self.k_w
    r1 = np.array(self.r1[-kw:])
    r2 = np.array(self.r2[-kw:])
    z = np.array(self.z_smoothed[-kw:])
    eta = np.array(self.eta_smoothed[-kw:])
    l1, l2 = np.log(r1), np.log(r2)
    for _ in range(self.outer_iter):                      # :362
      rho, mu = self.rho_d, self.mu_d
      p11 = (1.0 / r1) * np.sqrt(rho[0]) * np.exp(-0.5 * rho[0] * (l1 - mu[0]) ** 2)
      p12 = (1.0 / r1) * np.sqrt(rho[1]) * np.exp(-0.5 * rho[1] * (l1 - mu[1]) ** 2)
      p21 = (1.0 / r2) * np.sqrt(rho[1]) * np.exp(-0.5 * rho[1] * (l2 - mu[1]) ** 2)
      p22 = (1.0 / r2) * np.sqrt(rho[0]) * np.exp(-0.5 * rho[0] * (l2 - mu[0]) ** 2)
      p = 1.0 / (1.0 + np.exp(-z))                        # :376
      ep = (p * p11 * p21) / (p * p11 * p21 + (1.0 - p) * p12 * p22)   # :378
      self.mu_d[0] = (np.sum(ep * l1 + (1.0 - ep) * l2) +              # :381-385
                      kw * self.mu_0[0]) / (2.0 * kw)
      self.mu_d[1] = (np.sum(ep * l2 + (1.0 - ep) * l1) +
                      kw * self.mu_0[1]) / (2.0 * kw)
      self.rho_d[0] = (2.0 * kw * self.alpha_0[0]) / (                 # :387-395
          np.sum(ep * ((l1 - self.mu_d[0]) ** 2) +
                 (1.0 - ep) * ((l2 - self.mu_d[0]) ** 2)) +
          kw * (2.0 * self.beta_0[0] + (self.mu_d[0] - self.mu_0[0]) ** 2))
      self.rho_d[1] = (2.0 * kw * self.alpha_0[1]) / (
          np.sum(ep * ((l2 - self.mu_d[1]) ** 2) +
                 (1.0 - ep) * ((l1 - self.mu_d[1]) ** 2)) +
          kw * (2.0 * self.beta_0[1] + (self.mu_d[1] - self.mu_0[1]) ** 2))
      for _ in range(self.inner_iter):                    # :398
        for k in range(1, kw + 1):                        # :400-416 filter
          self.z_pred[k] = self.lam * self.z_kk[k - 1]
          self.s_pred[k] = self.lam ** 2 * self.s_kk[k - 1] + eta[k - 1]
          for _ in range(self.newton_iter):
            ez = np.exp(self.z_kk[k])
            self.z_kk[k] = self.z_kk[k] - (
                self.z_kk[k] - self.z_pred[k] -
                self.s_pred[k] * (ep[k - 1] - ez / (1 + ez))) / (
                    1 + self.s_pred[k] * ez / ((1 + ez) ** 2))
          ez = np.exp(self.z_kk[k])
          self.s_kk[k] = 1.0 / (1.0 / self.s_pred[k] + ez / ((1 + ez) ** 2))
        self.z_cap[kw] = self.z_kk[kw]                    # :419-430 smoother
        self.s_cap[kw] = self.s_kk[kw]
        for k in range(kw):                               # ascending k (sic)
          self.sm[k] = self.s_kk[k] * self.lam / self.s_pred[k + 1]
          self.z_cap[k] = self.z_kk[k] + self.sm[k] * (self.z_cap[k + 1] -
                                                       self.z_pred[k + 1])
          self.s_cap[k] = self.s_kk[k] + self.sm[k] ** 2 * (
              self.s_cap[k + 1] - self.s_pred[k + 1])
        self.z_kk[0] = self.z_cap[0]                      # :432-433
        self.s_kk[0] = self.s_cap[0]
        eta = ((self.z_cap[1:] - self.z_cap[:-1]) ** 2 +  # :435-437
               self.s_cap[1:] + self.s_cap[:-1] -
               2.0 * self.s_cap[1:] * self.sm + 2 * self.b_0) / (
                   1 + 2 * (self.a_0 + 1))
      z = self.z_cap[1:]                                  # :439 (a view, as there)
    self.z_smoothed += list(self.z_cap[1:])               # :442-446
    self.eta_smoothed += list(eta)
    self.z_kk[0] = self.z_cap[1]
    self.z_dyn.append(self.z_smoothed[-1 - self.k_f])
    self.eta_dyn.append(self.eta_smoothed[-1 - self.k_f])
    zd, ed = self.z_dyn[-1], self.eta_dyn[-1]
    return (1.0 / (1 + np.exp(-zd)),                      # :448-450
            1.0 / (1 + np.exp(-zd - self.c0 * np.sqrt(ed))),
            1.0 / (1 + np.exp(-zd + self.c0 * np.sqrt(ed))))


def decode_accuracy(attention_first_col, labels):
  """infer.py:406-407: correct = xor(attention >= 0.5, label) (attention True
  means speaker 1, i.e. label 0)."""
  att = np.asarray(attention_first_col, np.float64).reshape(-1, 1)
  labels = np.asarray(labels).reshape(-1, 1)
  correct = np.logical_xor(att >= 0.5, labels)
  return np.sum(correct) / float(len(correct))


def find_first_segment(labels):
  """infer.py:314-324."""
  labels = np.asarray(labels)
  end = np.nonzero(np.logical_xor(labels, labels[0]))
  if end[0].shape[0]:
    return int(end[0][0])
  return 0
