"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's skeletal feature extraction
(skeletal_network/skeletal_feature_extraction.py).  Parity unpinned by the reference (no fixtures exist); the
arithmetic is plain numpy fp64 exactly as the reference spells it (squares via ``**2``, ``sum(axis=0)``, ``np.sqrt``,
``np.arctan2``), on column arrays instead of a pandas frame.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np

JOINT_COLS = ['lhX', 'lhY', 'rhX', 'rhY', 'leX', 'leY', 'reX', 'reY', 'hipX', 'hipY', 'shcX', 'shcY']
FEATURE_COLS = ['lh_v', 'rh_v', 'le_v', 're_v', 'lh_a', 'rh_a', 'le_a', 're_a', 'hands_d',
                'lh_hip_d', 'rh_hip_d', 'le_hip_d', 're_hip_d', 'lh_shc_d', 'rh_shc_d', 'le_shc_d', 're_shc_d',
                'lh_hip_ang', 'rh_hip_ang', 'lh_shc_ang', 'rh_shc_ang', 'lh_el_ang', 'rh_el_ang']


def _prev(x):
    """get_previous_pos / get_previous_vel (:24-62): shift down by one row over the WHOLE table, first row 0."""
    p = np.zeros_like(x)
    p[1:] = x[:-1]
    return p


def _euclid(a, b):
    """(:86-93, :149-173): ``((a - b)**2).sum(axis=0)`` then sqrt, a and b stacked as (2, N)."""
    d = (a - b) ** 2
    return np.sqrt(d.sum(axis=0))


def extract_features(joints):
    """joints: dict column-name -> (N,) float64 array (the 12 JOINT_COLS).  Returns dict of the 23 FEATURE_COLS."""
    c = {k: np.asarray(joints[k], np.float64) for k in JOINT_COLS}
    pos = {j: np.array((c[j + 'X'], c[j + 'Y'])) for j in ('lh', 'rh', 'le', 're', 'hip', 'shc')}
    out = {}
    vel = {}
    for j in ('lh', 'rh', 'le', 're'):
        pre = np.array((_prev(c[j + 'X']), _prev(c[j + 'Y'])))
        dist = _euclid(pos[j], pre)
        v = np.zeros_like(c[j + 'X'])
        v[5:] = dist[5:]                       # calculate_velocities (:98-99)
        vel[j] = v
        out[j + '_v'] = v
    for j in ('lh', 'rh', 'le', 're'):
        dv = vel[j] - _prev(vel[j])            # calculate_accelerations (:118-126)
        a = np.zeros_like(dv)
        a[5:] = dv[5:]
        out[j + '_a'] = a
    out['hands_d'] = _euclid(pos['lh'], pos['rh'])            # calculate_distances (:148-184)
    for ref in ('hip', 'shc'):
        for j in ('lh', 'rh', 'le', 're'):
            out['%s_%s_d' % (j, ref)] = _euclid(pos[j], pos[ref])
    for ref, tag in (('hip', 'hip'), ('shc', 'shc')):         # calculate_angles (:198-213)
        for j in ('lh', 'rh'):
            d = pos[j] - pos[ref]
            out['%s_%s_ang' % (j, tag)] = np.arctan2(d[1], d[0])
    for j, e in (('lh', 'le'), ('rh', 're')):
        d = pos[j] - pos[e]
        out['%s_el_ang' % j] = np.arctan2(d[1], d[0])
    return {k: out[k] for k in FEATURE_COLS}
