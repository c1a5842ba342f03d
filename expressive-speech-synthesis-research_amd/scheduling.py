"""Training schedules (TransformerTTS/utils/scheduling.py:5-47): piecewise-linear values and step-wise reductions read
from the `*_schedule` lists of the yaml configs ([[step, value], ...])."""
import numpy as np


def piecewise_linear(step, X, Y):
    """Value at `step` of the piecewise-linear function through (X_i, Y_i); constant outside the breakpoints."""
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    if len(X) != len(Y):
        raise ValueError('breakpoints and values differ in length')
    if step < X[0]:
        return float(Y[0])
    idx = int(np.where(step >= X)[0][-1])
    if idx == len(Y) - 1:
        return float(Y[-1])
    slope = (Y[idx + 1] - Y[idx]) / (X[idx + 1] - X[idx])
    return float(Y[idx] + slope * (step - X[idx]))


def piecewise_linear_schedule(step, schedule):
    s = np.asarray(schedule, dtype=np.float64)
    return np.float32(piecewise_linear(step, s[:, 0], s[:, 1]))


def reduction_schedule(step, schedule):
    """The value of the last entry whose step has been reached (reduction factor, heads to drop)."""
    s = np.asarray(schedule)
    r = s[0, 0]
    for i in range(s.shape[0]):
        if s[i, 0] <= step:
            r = s[i, 1]
        else:
            break
    return int(r)
