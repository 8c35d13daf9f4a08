#!/usr/bin/env python3
"""Fixture: the numbers of the reference's deployment mapping (policy command -> Ackermann drive message), read out of the
reference's ROS nodes in the build container and committed as DATA (tests/golden/deployment_mapping.json).

    python tests/golden/make_golden_deployment_mapping.py        (needs /root/reference; the tests only read the .json)

Every number is found by a pattern in the cited line; the script fails if a pattern no longer matches."""
import hashlib
import json
import os
import re

REF = "/root/reference/ros_agent"
HERE = os.path.dirname(os.path.abspath(__file__))


def grab(path, pattern, cast=float, flags=0):
    text = open(os.path.join(REF, path)).read()
    m = re.search(pattern, text, flags)
    assert m, (path, pattern)
    line = text[:m.start()].count("\n") + 1
    return [cast(g) for g in m.groups()], line, hashlib.sha256(text.encode()).hexdigest()[:16]


def main():
    out = {"_made_by": "tests/golden/make_golden_deployment_mapping.py", "_sources": {}}
    d, a = "agents/dreamer/src/agent.py", {}
    (k_hw,), ln, sha = grab(d, r"^\s*steering = 0 - float\(action\['steering'\] \* ([0-9.]+) \* 0\.42\) # working better in hardware", flags=re.M)
    a.update(scale_hardware=k_hw, line_hardware=ln)
    out["_sources"][d] = sha
    (k_sim,), ln, _ = grab(d, r"^\s*# steering = 0 - float\(action\['steering'\] \* ([0-9.]+) \* 0\.42\) # working better in simulation", flags=re.M)
    a.update(scale_simulation=k_sim, line_simulation=ln)
    (n1, d1, n2, d2), ln, _ = grab(d, r"self\._steering = float\(self\._steering\) \* (\d+) / (\d+) \+ float\(steering\) \* (\d+) / (\d+) # lowpass in simulation")
    a.update(lowpass_simulation=[n1 / d1, n2 / d2], line_lowpass_simulation=ln)
    (period, slack), ln, _ = grab(d, r"since_last_laserscan\.to_sec\(\) < \(([0-9.]+) - ([0-9.]+)\): # limit to approx\. 10Hz")
    a.update(min_decision_period_s=period - slack, line_rate=ln)
    (thr,), ln, _ = grab(d, r"if float\(action\['motor'\]\) < ([0-9.]+):")
    a.update(motor_threshold=thr, line_motor=ln)
    (dn,), _, _ = grab(d, r"self\._motor = self\._motor - float\(self\._config_b\)/1000 #([0-9.]+)")
    (up,), _, _ = grab(d, r"self\._motor = self\._motor \+ float\(self\._config_a\)/1000 #([0-9.]+)")
    (hi,), _, _ = grab(d, r"if self\._motor > ([0-9.]+):")
    (lo,), _, _ = grab(d, r"if self\._motor < ([0-9.]+):")
    a.update(speed_step_up=up, speed_step_down=dn, speed_clip=[lo, hi])
    # the sign: "0 - float(action['steering'] ..." published as drive.steering_angle, ROS convention positive = left
    a["sign"] = -1
    a["steering_angle_convention"] = "ackermann_msgs/AckermannDrive.steering_angle: positive = left (counter-clockwise), REP 103"
    out["dreamer_node"] = a
    for node in ("acme", "sb3"):
        p = f"agents/{node}/src/agent.py"
        (k,), ln, sha = grab(p, r"^\s*steering = 0 - float\(action\['steering'\] \* ([0-9.]+) \* 0\.42\) # working better in hardware", flags=re.M)
        (lo,), _, _ = grab(p, r"if self\._motor < ([0-9.]+):")
        (hi,), _, _ = grab(p, r"if self\._motor > ([0-9.]+):")
        out[f"{node}_node"] = {"scale": k, "line": ln, "sign": -1, "speed_clip": [lo, hi]}
        out["_sources"][p] = sha
    (nom,), ln, sha = grab("models/dreamer/racing_dreamer.py", r"self\._max_steering_angle = ([0-9.]+)")
    out["nominal_max_steering_angle"] = nom
    out["_sources"]["models/dreamer/racing_dreamer.py"] = sha
    ks = [out["dreamer_node"]["scale_hardware"], out["dreamer_node"]["scale_simulation"], out["acme_node"]["scale"], out["sb3_node"]["scale"]]
    out["effective_lock_band_rad"] = [min(ks) * nom, max(ks) * nom]
    with open(os.path.join(HERE, "deployment_mapping.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
