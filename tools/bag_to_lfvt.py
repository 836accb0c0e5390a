#!/usr/bin/env python3
"""rosbag -> LFVT (lf-vio_amd/host/replay.h): the one command between a recorded run of the reference and
`lfvio_host_replay` / tools/replay_stream.py / tools/ate.py — BASELINE configs[2] (PALVIO ID01) once the bag exists.

    rosbag record -O id01_topics.bag /imu0 /feature_tracker/feature /feature_tracker/restart /vins_estimator/lfvt_bootstrap
    python tools/bag_to_lfvt.py id01_topics.bag id01.lfvt [--imu /imu0] [--truth gt.txt]

The first three topics are the node's own inputs (estimator_node.cpp:352-356); the fourth is the dump hook of
INTEGRATION.md section 3 (one std_msgs/Float64MultiArray per successful initialStructure()).  Messages are written in the
order the bag received them, contents unchanged (float32 bearings and channels stay float32).  --truth adds ground-truth
records from a `stamp x y z qx qy qz qw` text file for tools/ate.py.  No ROS installation is needed."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))


def convert(bag_path, out_path, imu_topic="/imu0", feature_topic="/feature_tracker/feature", restart_topic="/feature_tracker/restart",
            bootstrap_topic="/vins_estimator/lfvt_bootstrap", truth_path=None):
    import numpy as np
    from lfvio import rosmsg, trace

    w = trace.TraceWriter(out_path)
    n = dict(imu=0, images=0, restarts=0, bootstraps=0, other=0)
    for topic, mtype, t, payload in rosmsg.read_bag(bag_path):
        if topic == imu_topic:
            stamp, acc, gyr = rosmsg.de_imu(payload)
            w.imu(stamp, acc, gyr)
            n["imu"] += 1
        elif topic == feature_topic:
            stamp, rec = rosmsg.de_pointcloud(payload)
            w._rec(trace.REC_FEATURES, __import__("struct").pack("<dI", stamp, len(rec)) + rec.tobytes())
            n["images"] += 1
        elif topic == restart_topic:
            if rosmsg.de_bool(payload):
                w.restart(t)
                n["restarts"] += 1
        elif topic == bootstrap_topic:
            w.bootstrap_payload(rosmsg.de_f64_array(payload))
            n["bootstraps"] += 1
        else:
            n["other"] += 1
    if truth_path:
        for row in np.loadtxt(truth_path).reshape(-1, 8):
            w.truth(row[0], row[1:4], row[4:8])
    w.close()
    return n


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("bag")
    ap.add_argument("out")
    ap.add_argument("--imu", default="/imu0")
    ap.add_argument("--features", default="/feature_tracker/feature")
    ap.add_argument("--restart", default="/feature_tracker/restart")
    ap.add_argument("--bootstrap", default="/vins_estimator/lfvt_bootstrap")
    ap.add_argument("--truth", default=None)
    a = ap.parse_args()
    print(convert(a.bag, a.out, a.imu, a.features, a.restart, a.bootstrap, a.truth))
