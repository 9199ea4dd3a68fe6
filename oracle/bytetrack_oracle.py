"""CPU restatement of the reference's ByteTrack association (row N2, second half): unicorn/tracker/byte_tracker.py,
matching.py (iou_distance, fuse_score, linear_assignment), kalman_filter.py, basetrack.py.

TEST INFRASTRUCTURE ONLY.  numpy float64 like the reference; every function cites the lines it follows.
Third-party arithmetic that is absent offline and therefore restated from its published semantics ("parity unpinned" for
these two):
  * cython_bbox.bbox_overlaps (requirements.txt, unpinned): Fast R-CNN IoU with the +1 pixel convention;
  * lap.lapjv(cost, extend_cost=True, cost_limit=t) (requirements.txt:28, unpinned): the cost matrix is embedded in an
    (n_rows+n_cols)^2 square whose off blocks cost t/2 and whose lower-right block costs 0, solved exactly; rows/columns
    assigned into the extension are unmatched.  Any exact solver gives the same real matches unless real costs tie.
Everything else is pinned: tests/golden/byte_sequence.npz was produced by the REAL reference classes (with the two stubs
above installed by oracle/ref_bootstrap.py) and tests/test_assoc_cpu.py holds this restatement to it.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3   # basetrack.py:5-9


def bbox_overlaps(a, b):
    """cython_bbox.bbox_overlaps: IoU of (N,4) x (K,4) xyxy boxes, +1 convention, float64"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    out = np.zeros((a.shape[0], b.shape[0]))
    for k in range(b.shape[0]):
        box_area = (b[k, 2] - b[k, 0] + 1) * (b[k, 3] - b[k, 1] + 1)
        for n in range(a.shape[0]):
            iw = min(a[n, 2], b[k, 2]) - max(a[n, 0], b[k, 0]) + 1
            if iw > 0:
                ih = min(a[n, 3], b[k, 3]) - max(a[n, 1], b[k, 1]) + 1
                if ih > 0:
                    ua = (a[n, 2] - a[n, 0] + 1) * (a[n, 3] - a[n, 1] + 1) + box_area - iw * ih
                    out[n, k] = iw * ih / ua
    return out


def lapjv(cost, extend_cost=True, cost_limit=np.inf):
    """lap.lapjv semantics used by matching.py:43: returns (opt, x, y), x[i] = column of row i or -1"""
    cost = np.asarray(cost, np.float64)
    nr, nc = cost.shape
    n = nr + nc
    ext = np.full((n, n), cost_limit / 2.0)
    ext[nr:, nc:] = 0
    ext[:nr, :nc] = cost
    r, c = linear_sum_assignment(ext)
    x = np.full(nr, -1, dtype=int)
    y = np.full(nc, -1, dtype=int)
    for i, j in zip(r, c):
        if i < nr and j < nc:
            x[i], y[j] = j, i
    return float(ext[r, c].sum()), x, y


def linear_assignment(cost_matrix, thresh):
    """matching.py:39-51"""
    if cost_matrix.size == 0:
        return np.empty((0, 2), dtype=int), tuple(range(cost_matrix.shape[0])), tuple(range(cost_matrix.shape[1]))
    _, x, y = lapjv(cost_matrix, extend_cost=True, cost_limit=thresh)
    matches = np.asarray([[ix, mx] for ix, mx in enumerate(x) if mx >= 0])
    return matches, np.where(x < 0)[0], np.where(y < 0)[0]


class KalmanFilter:
    """kalman_filter.py:40-225 (initiate, multi_predict, project, update)"""

    def __init__(self):
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)
        self.wp, self.wv = 1.0 / 20, 1.0 / 160

    def initiate(self, m):
        mean = np.r_[m, np.zeros_like(m)]
        std = [2 * self.wp * m[3], 2 * self.wp * m[3], 1e-2, 2 * self.wp * m[3],
               10 * self.wv * m[3], 10 * self.wv * m[3], 1e-5, 10 * self.wv * m[3]]
        return mean, np.diag(np.square(std))

    def predict(self, mean, cov):      # one row of multi_predict (:110-142); note the 1e-2 / 1e-5 aspect-ratio terms
        std = [self.wp * mean[3], self.wp * mean[3], 1e-2, self.wp * mean[3],
               self.wv * mean[3], self.wv * mean[3], 1e-5, self.wv * mean[3]]
        return self.F @ mean, self.F @ cov @ self.F.T + np.diag(np.square(std))

    def project(self, mean, cov):
        std = [self.wp * mean[3], self.wp * mean[3], 1e-1, self.wp * mean[3]]
        return self.H @ mean, self.H @ cov @ self.H.T + np.diag(np.square(std))

    def update(self, mean, cov, z):
        pm, pc = self.project(mean, cov)
        K = np.linalg.solve(pc, (cov @ self.H.T).T).T          # cho_factor / cho_solve (:215-219)
        return mean + (z - pm) @ K.T, cov - K @ pc @ K.T


class Track:
    """byte_tracker.py:13-140 (STrack) without the python object plumbing"""

    def __init__(self, tlwh, score):
        self._tlwh = np.asarray(tlwh, dtype=np.float64)
        self.mean = self.cov = None
        self.is_activated = False
        self.score = score
        self.tracklet_len = 0
        self.state = NEW
        self.track_id = 0
        self.frame_id = self.start_frame = 0

    @property
    def tlwh(self):
        if self.mean is None:
            return self._tlwh.copy()
        r = self.mean[:4].copy()
        r[2] *= r[3]
        r[:2] -= r[2:] / 2
        return r

    @property
    def tlbr(self):
        r = self.tlwh.copy()
        r[2:] += r[:2]
        return r


def xyah(tlwh):
    r = np.asarray(tlwh).copy()
    r[:2] += r[2:] / 2
    r[2] /= r[3]
    return r


def iou_distance(a, b):
    """matching.py:75-91"""
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)))
    return 1 - bbox_overlaps([t.tlbr for t in a], [t.tlbr for t in b])


def fuse_score(cost, dets):
    """matching.py:173-181"""
    if cost.size == 0:
        return cost
    s = np.array([d.score for d in dets])
    return 1 - (1 - cost) * np.expand_dims(s, 0).repeat(cost.shape[0], axis=0)


class ByteState:
    def __init__(self, track_thresh=0.6, track_buffer=30, match_thresh=0.9, mot20=False, frame_rate=30):
        self.track_thresh, self.match_thresh, self.mot20 = track_thresh, match_thresh, mot20
        self.det_thresh = track_thresh + 0.1                       # byte_tracker.py:152
        self.max_time_lost = int(frame_rate / 30.0 * track_buffer)
        self.tracked, self.lost, self.removed_ids = [], [], set()
        self.frame_id = 0
        self.kf = KalmanFilter()
        self.count = 0                                             # BaseTrack._count (per state here)

    def next_id(self):
        self.count += 1
        return self.count


def _joint(a, b):
    seen, res = set(), []
    for t in list(a) + list(b):
        if t.track_id not in seen:
            seen.add(t.track_id)
            res.append(t)
    return res


def byte_update(st: ByteState, output_results, img_info, img_size):
    """byte_tracker.py:156-293.  output_results float32 (n,5) [x1,y1,x2,y2,score] or (n,>=6) (score = c4*c5)"""
    st.frame_id += 1
    activated, refind, lost_new, removed_new = [], [], [], []
    out = np.array(output_results, dtype=np.float32, copy=True)
    scores = out[:, 4] if out.shape[1] == 5 else out[:, 4] * out[:, 5]
    bboxes = out[:, :4]
    scale = min(img_size[0] / float(img_info[0]), img_size[1] / float(img_info[1]))
    bboxes = bboxes / scale                                        # float32 array / python float -> float32 (:172)
    remain = scores > st.track_thresh
    second = np.logical_and(scores > 0.1, scores < st.track_thresh)
    dets = [Track([b[0], b[1], b[2] - b[0], b[3] - b[1]], s) for b, s in zip(bboxes[remain], scores[remain])]
    dets2 = [Track([b[0], b[1], b[2] - b[0], b[3] - b[1]], s) for b, s in zip(bboxes[second], scores[second])]
    unconfirmed = [t for t in st.tracked if not t.is_activated]
    tracked = [t for t in st.tracked if t.is_activated]
    pool = _joint(tracked, st.lost)
    for t in pool:                                                 # STrack.multi_predict (:33-45)
        m = t.mean.copy()
        if t.state != TRACKED:
            m[7] = 0
        t.mean, t.cov = st.kf.predict(m, t.cov)
    d = iou_distance(pool, dets)
    if not st.mot20:
        d = fuse_score(d, dets)
    matches, u_track, u_det = linear_assignment(d, st.match_thresh)

    def hit(t, det, fresh_list, refind_list):
        t.mean, t.cov = st.kf.update(t.mean, t.cov, xyah(det.tlwh))
        if t.state == TRACKED:                                     # update (:75-91)
            t.tracklet_len += 1
            fresh_list.append(t)
        else:                                                      # re_activate (:61-73)
            t.tracklet_len = 0
            refind_list.append(t)
        t.frame_id = st.frame_id
        t.state, t.is_activated, t.score = TRACKED, True, det.score

    for it, idet in matches:
        hit(pool[it], dets[idet], activated, refind)
    r_tracked = [pool[i] for i in u_track if pool[i].state == TRACKED]
    d = iou_distance(r_tracked, dets2)
    matches, u_track2, _ = linear_assignment(d, 0.5)
    for it, idet in matches:
        hit(r_tracked[it], dets2[idet], activated, refind)
    for it in u_track2:
        t = r_tracked[it]
        if t.state != LOST:
            t.state = LOST
            lost_new.append(t)
    dets = [dets[i] for i in u_det]
    d = iou_distance(unconfirmed, dets)
    if not st.mot20:
        d = fuse_score(d, dets)
    matches, u_unc, u_det = linear_assignment(d, 0.7)
    for it, idet in matches:
        hit(unconfirmed[it], dets[idet], activated, refind)        # state is TRACKED for unconfirmed tracks -> update()
    for it in u_unc:
        unconfirmed[it].state = REMOVED
        removed_new.append(unconfirmed[it])
    for i in u_det:                                                # Step 4 (:256-262) / activate (:47-59)
        t = dets[i]
        if t.score < st.det_thresh:
            continue
        t.track_id = st.next_id()
        t.mean, t.cov = st.kf.initiate(xyah(t._tlwh))
        t.tracklet_len, t.state = 0, TRACKED
        t.is_activated = st.frame_id == 1
        t.frame_id = t.start_frame = st.frame_id
        activated.append(t)
    for t in st.lost:                                              # Step 5 (:263-267)
        if st.frame_id - t.frame_id > st.max_time_lost:
            t.state = REMOVED
            removed_new.append(t)
    st.tracked = _joint(_joint([t for t in st.tracked if t.state == TRACKED], activated), refind)
    ids = {t.track_id for t in st.tracked}
    st.lost = [t for t in {t.track_id: t for t in st.lost}.values() if t.track_id not in ids]     # sub_stracks (:308-317)
    st.lost.extend(lost_new)
    st.lost = [t for t in {t.track_id: t for t in st.lost}.values() if t.track_id not in st.removed_ids]
    st.removed_ids.update(t.track_id for t in removed_new)
    # remove_duplicate_stracks (:320-337)
    pd = iou_distance(st.tracked, st.lost)
    dupa, dupb = set(), set()
    for p, q in zip(*np.where(pd < 0.15)):
        tp = st.tracked[p].frame_id - st.tracked[p].start_frame
        tq = st.lost[q].frame_id - st.lost[q].start_frame
        (dupb.add(q) if tp > tq else dupa.add(p))
    st.tracked = [t for i, t in enumerate(st.tracked) if i not in dupa]
    st.lost = [t for i, t in enumerate(st.lost) if i not in dupb]
    return [t for t in st.tracked if t.is_activated]


def synth_detections(n_frames=60, n_obj=12, seed=0, H=1080, W=1920, size=(800, 1440)):
    """Deterministic MOT-style detection stream in NETWORK coordinates (the tracker divides by the letterbox scale):
    linear motion + jitter, score drift, occlusions (missed / low-score detections), crossings, false positives."""
    g = np.random.default_rng(seed)
    scale = min(size[0] / H, size[1] / W)
    pos = g.uniform([100, 100], [W - 300, H - 300], (n_obj, 2))
    wh = g.uniform([40, 90], [120, 260], (n_obj, 2))
    vel = g.uniform(-9, 9, (n_obj, 2))
    born = np.sort(g.integers(0, n_frames // 2, n_obj)); born[:5] = 0
    frames = []
    for f in range(n_frames):
        pos = pos + vel + g.normal(0, 1.2, pos.shape)
        rows = []
        for o in range(n_obj):
            if f < born[o]:
                continue
            u = g.random()
            if u < 0.08:
                continue                                           # missed
            sc = g.uniform(0.65, 0.98) if u > 0.22 else g.uniform(0.15, 0.55)    # occluded -> low score (second association)
            b = np.r_[pos[o], pos[o] + wh[o]] + g.normal(0, 1.5, 4)
            rows.append(np.r_[b * scale, sc])
        for _ in range(g.integers(0, 3)):                          # false positives
            p = g.uniform([0, 0], [W - 100, H - 100]); s = g.uniform(20, 80, 2)
            rows.append(np.r_[np.r_[p, p + s] * scale, g.uniform(0.12, 0.75)])
        frames.append(np.asarray(rows, dtype=np.float32).reshape(-1, 5))
    return frames, (H, W), size
