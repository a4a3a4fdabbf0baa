"""Skeletal feature extraction on the GPU (reference skeletal_network/skeletal_feature_extraction.py:16-215).

Same function names and column names as the reference; every function takes the frame table (a pandas DataFrame or
a dict of equal-length column arrays holding at least lhX..shcY) and returns it with the new columns added.  The
arithmetic of all of them is ONE HIP kernel over the whole table (``mgr_skeletal_features``, fp64); the individual
functions pick their columns out of its result, so calling them in the reference's order
(get_previous_pos -> calculate_velocities -> get_previous_vel -> calculate_accelerations -> calculate_distances ->
calculate_angles) costs one launch, cached on the table's identity.
"""
import numpy as np

from .. import _capi

JOINT_COLS = ['lhX', 'lhY', 'rhX', 'rhY', 'leX', 'leY', 'reX', 'reY', 'hipX', 'hipY', 'shcX', 'shcY']
FEATURE_COLS = ['lh_v', 'rh_v', 'le_v', 're_v', 'lh_a', 'rh_a', 'le_a', 're_a', 'hands_d',
                'lh_hip_d', 'rh_hip_d', 'le_hip_d', 're_hip_d', 'lh_shc_d', 'rh_shc_d', 'le_shc_d', 're_shc_d',
                'lh_hip_ang', 'rh_hip_ang', 'lh_shc_ang', 'rh_shc_ang', 'lh_el_ang', 'rh_el_ang']
_DEV = [None]
_CACHE = {}


def _device():
    if _DEV[0] is None:
        _DEV[0] = _capi.Device(0)
    return _DEV[0]


def load_data(sk_data_file):
    """(:16-20) the whole training set frame by frame."""
    import pandas as pd
    return pd.read_csv(sk_data_file)


def features_array(joints, dev=None):
    """joints (N,12) float64 in JOINT_COLS order -> (N,23) float64 in FEATURE_COLS order, computed on the GPU."""
    dev = dev or _device()
    J = np.ascontiguousarray(joints, dtype=np.float64)
    if J.ndim != 2 or J.shape[1] != len(JOINT_COLS):
        raise ValueError("joints must be (N, 12)")
    n = J.shape[0]
    out = np.empty((n, len(FEATURE_COLS)), np.float64)
    if n == 0:
        return out
    dJ = dev.array(J)
    dO = dev.empty((n, len(FEATURE_COLS)), np.float64)
    dev.call("mgr_skeletal_features", dJ, n, dO)
    out = dO.download()
    dJ.free()
    dO.free()
    return out


def _features(df):
    key = id(df)
    hit = _CACHE.get(key)
    n = len(df[JOINT_COLS[0]])
    if hit is not None and hit[0] == n:
        return hit[1]
    J = np.stack([np.asarray(df[c], np.float64) for c in JOINT_COLS], axis=1)
    F = features_array(J)
    _CACHE.clear()
    _CACHE[key] = (n, F)
    return F


def _put(df, names):
    F = _features(df)
    for nme in names:
        df[nme] = F[:, FEATURE_COLS.index(nme)]
    return df


def _shift(x):
    x = np.asarray(x)
    p = np.zeros_like(x)
    p[1:] = x[:-1]
    return p


def get_previous_pos(df):
    """(:24-43) adds pre_lhX .. pre_reY: the previous row's hand / elbow positions (plain data movement, host side)."""
    for j in ('lh', 'rh', 'le', 're'):
        for ax in ('X', 'Y'):
            df['pre_' + j + ax] = _shift(df[j + ax])
    return df


def calculate_velocities(df):
    """(:70-104) lh_v rh_v le_v re_v."""
    return _put(df, FEATURE_COLS[0:4])


def get_previous_vel(df):
    """(:47-64) pre_lh_v .. pre_re_v."""
    for j in ('lh', 'rh', 'le', 're'):
        df['pre_' + j + '_v'] = _shift(df[j + '_v'])
    return df


def calculate_accelerations(df):
    """(:108-130) lh_a rh_a le_a re_a."""
    return _put(df, FEATURE_COLS[4:8])


def calculate_distances(df):
    """(:135-186) hands_d and the 8 hand/elbow - hip / shoulder-centre distances."""
    return _put(df, FEATURE_COLS[8:17])


def calculate_angles(df):
    """(:191-215) the 6 hand - hip / shoulder-centre / elbow angles."""
    return _put(df, FEATURE_COLS[17:23])


def extract_features(df):
    """The reference's main sequence (:297-309) in one call."""
    df = get_previous_pos(df)
    df = calculate_velocities(df)
    df = get_previous_vel(df)
    df = calculate_accelerations(df)
    df = calculate_distances(df)
    return calculate_angles(df)
