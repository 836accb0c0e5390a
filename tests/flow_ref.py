"""Plain-Python restatement of the control flow around optimization() (test infrastructure):
  FeatureManager::addFeatureCheckParallax / compensatedParallax2   feature_manager.cpp:45-95, 353-369
  Estimator::processIMU, processImage (INITIAL branch), slideWindow, removeBack / removeFront
                                                                 estimator.cpp:86-220, 1011-1131; feature_manager.cpp:312-351
  getMeasurements() + the IMU loop of process()                  estimator_node.cpp:96-134, 218-262
Only what runs WITHOUT the solver: the window bookkeeping before initialization and the message synchronisation."""
import numpy as np

W = 10  # WINDOW_SIZE


def delta_R(theta):
    """Utility::deltaQ(theta).toRotationMatrix(): quaternion (1, theta/2), NOT normalised (utility.h:20-30)"""
    w, x, y, z = 1.0, theta[0] / 2, theta[1] / 2, theta[2] / 2
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


class Flow:
    def __init__(self, min_parallax, g=(0, 0, 0)):
        self.min_parallax = min_parallax
        self.g = np.array(g, dtype=float)
        self.frame_count = 0
        self.first_imu = False
        self.Ps, self.Vs = np.zeros((W + 1, 3)), np.zeros((W + 1, 3))
        self.Rs = np.tile(np.eye(3), (W + 1, 1, 1))
        self.Bas, self.Bgs = np.zeros((W + 1, 3)), np.zeros((W + 1, 3))
        self.Headers = np.zeros(W + 1)
        self.bufs = [[] for _ in range(W + 1)]          # per frame: list of (dt, acc, gyr)
        self.has_pre = [False] * (W + 1)
        self.feature = []                                # [id, start_frame, [points...]] in list order
        self.last_track_num = 0
        self.sum_of_back = self.sum_of_front = 0
        self.marg_old = True
        self.acc_0 = self.gyr_0 = None

    # estimator.cpp:86-120
    def process_imu(self, dt, acc, gyr):
        acc, gyr = np.asarray(acc, float), np.asarray(gyr, float)
        if not self.first_imu:
            self.first_imu, self.acc_0, self.gyr_0 = True, acc, gyr
        j = self.frame_count
        self.has_pre[j] = True
        if j != 0:
            self.bufs[j].append((dt, acc, gyr))
            un_acc_0 = self.Rs[j] @ (self.acc_0 - self.Bas[j]) - self.g
            un_gyr = 0.5 * (self.gyr_0 + gyr) - self.Bgs[j]
            self.Rs[j] = self.Rs[j] @ delta_R(un_gyr * dt)
            un_acc_1 = self.Rs[j] @ (acc - self.Bas[j]) - self.g
            un_acc = 0.5 * (un_acc_0 + un_acc_1)
            self.Ps[j] = self.Ps[j] + dt * self.Vs[j] + 0.5 * dt * dt * un_acc
            self.Vs[j] = self.Vs[j] + dt * un_acc
        self.acc_0, self.gyr_0 = acc, gyr

    # feature_manager.cpp:45-95
    def add_feature_check_parallax(self, frame_count, ids, pts):
        parallax_sum, parallax_num, self.last_track_num = 0.0, 0, 0
        for fid, p in sorted(zip(ids, pts), key=lambda a: a[0]):  # std::map iterates by id
            hit = [f for f in self.feature if f[0] == fid]
            if not hit:
                self.feature.append([fid, frame_count, [np.array(p[:3], float)]])
            else:
                hit[0][2].append(np.array(p[:3], float))
                self.last_track_num += 1
        if frame_count < 2 or self.last_track_num < 20:
            return True
        for fid, start, obs in self.feature:
            if start <= frame_count - 2 and start + len(obs) - 1 >= frame_count - 1:
                p_i, p_j = obs[frame_count - 2 - start], obs[frame_count - 1 - start]
                parallax_sum += np.arccos(p_i @ p_j) * 10  # :353-369
                parallax_num += 1
        if parallax_num == 0:
            return True
        return parallax_sum / parallax_num >= self.min_parallax

    # estimator.cpp:122-220, solver_flag == INITIAL and initialStructure() == false
    def process_image(self, ids, pts, stamp):
        self.marg_old = self.add_feature_check_parallax(self.frame_count, ids, pts)
        self.Headers[self.frame_count] = stamp
        if self.frame_count == W:
            self.slide_window()
        else:
            self.frame_count += 1

    # estimator.cpp:1011-1131 with solver_flag == INITIAL (removeBack, not removeBackShiftDepth)
    def slide_window(self):
        if self.frame_count != W:
            return
        if self.marg_old:
            for a in (self.Ps, self.Vs, self.Rs, self.Bas, self.Bgs, self.Headers):
                a[:-1] = a[1:].copy()  # the chain of swaps ends as a rotation; slot W is overwritten below
            self.bufs = self.bufs[1:] + [[]]
            self.has_pre = self.has_pre[1:] + [True]
            self.sum_of_back += 1
            keep = []
            for f in self.feature:  # removeBack, feature_manager.cpp:312-328
                if f[1] != 0:
                    f[1] -= 1
                    keep.append(f)
                else:
                    f[2].pop(0)
                    if f[2]:
                        keep.append(f)
            self.feature = keep
        else:
            self.bufs[W - 1] = self.bufs[W - 1] + self.bufs[W]
            for a in (self.Ps, self.Vs, self.Rs, self.Bas, self.Bgs, self.Headers):
                a[W - 1] = a[W]
            self.bufs[W] = []
            self.sum_of_front += 1
            keep = []
            for f in self.feature:  # removeFront(frame_count), feature_manager.cpp:330-351
                if f[1] == W:
                    f[1] -= 1
                    keep.append(f)
                else:
                    j = W - 1 - f[1]
                    if f[1] + len(f[2]) - 1 >= W - 1:
                        f[2].pop(j)
                    if f[2]:
                        keep.append(f)
            self.feature = keep


def sync(imu, images, td_of):
    """getMeasurements() + the IMU loop of process() over complete message lists: yields per accepted image
    (stamp, index, [(dt, acc, gyr), ...]) exactly as processIMU would be called.  imu: [n,7] (t, acc, gyr); td_of(): current td."""
    front, current_time = 0, -1.0
    d = np.zeros(6)
    for idx, (stamp, _) in enumerate(images):
        td = td_of()
        if front >= len(imu) or not (imu[-1, 0] > stamp + td):
            return
        if not (imu[front, 0] < stamp + td):
            continue
        first = front
        while imu[front, 0] < stamp + td:
            front += 1
        calls = []
        img_t = stamp + td
        for k in range(first, front + 1):
            t = imu[k, 0]
            if t <= img_t:
                if current_time < 0:
                    current_time = t
                dt = t - current_time
                current_time = t
                d = imu[k, 1:7].copy()
                calls.append((dt, d[:3].copy(), d[3:].copy()))
            else:
                dt_1, dt_2 = img_t - current_time, t - img_t
                current_time = img_t
                w1, w2 = dt_2 / (dt_1 + dt_2), dt_1 / (dt_1 + dt_2)
                d = w1 * d + w2 * imu[k, 1:7]
                calls.append((dt_1, d[:3].copy(), d[3:].copy()))
        yield stamp, idx, calls
