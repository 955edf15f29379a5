#!/usr/bin/env python3
"""Writes configs/<robot>/*.yaml in the reference's own configuration schema (task stacks as in
/root/reference/etc/<robot>/tasks.yaml, CONTROLLER / BEHAVIOR trees as in etc/talos/talos_pos_tracker.yaml and
etc/talos/squat.yaml) from the constants below (the same numbers inria_wbc_amd/structure.py carries).
The C++ facade parses these files."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TALOS_CONTACT = dict(kp=30.0, lxp=0.1, lxn=0.11, lyp=0.069, lyn=0.069, lz=0.107, fmin=5.0, fmax=1500.0, mu=0.3, normal="[0, 0, 1]")
# (name, type, fields) in task-stack order
TALOS = [
    ("head", "se3", dict(tracked="head_1_joint", weight=1.0, kp=1.0, mask="110000")),
    ("head_pitch", "se3", dict(tracked="head_1_joint", weight=100.0, kp=30.0, mask="000010")),
    ("head_yaw", "se3", dict(tracked="head_2_joint", weight=100.0, kp=30.0, mask="000001")),
    ("lh", "se3", dict(tracked="gripper_left_joint", weight=10.0, kp=30.0, mask="111111")),
    ("rh", "se3", dict(tracked="gripper_right_joint", weight=10.0, kp=30.0, mask="111111")),
    ("torso", "se3", dict(tracked="torso_2_link", weight=10.0, kp=30.0, mask="000110")),
    ("lf", "se3", dict(tracked="leg_left_6_joint", weight=1000.0, kp=30.0, mask="111111")),
    ("rf", "se3", dict(tracked="leg_right_6_joint", weight=1000.0, kp=30.0, mask="111111")),
    ("com", "com", dict(weight=1000.0, kp=30.0, mask="111")),
    ("posture", "posture", dict(weight=1.75, kp=10.0, ref="inria_start")),
    ("momentum", "momentum", dict(weight=1000.0, kp=30.0, mask="000110")),
    ("bounds", "bounds", dict(weight=10000)),
    ("actuation_bounds", "actuation-bounds", dict(weight=10000)),
    ("contact_lfoot", "contact", dict(joint="leg_left_6_joint", **TALOS_CONTACT)),
    ("contact_rfoot", "contact", dict(joint="leg_right_6_joint", **TALOS_CONTACT)),
] + [(n, "self-collision", dict(tracked=t, radius=r, weight=w, kp=kp, kd=250.0, margin=0.02, m=0.2)) for n, t, r, w, kp in [
    ("self_collision-left", "gripper_left_joint", 0.1, 2000, 50.0), ("self_collision-right", "gripper_right_joint", 0.1, 2000, 50.0),
    ("self_collision-elbow-right", "arm_right_4_joint", 0.15, 1000, 50.0), ("self_collision-elbow-left", "arm_left_4_joint", 0.15, 1000, 150.0),
    ("self_collision-wrist-right", "arm_right_5_joint", 0.10, 1000, 50.0), ("self_collision-wrist-left", "arm_left_5_joint", 0.10, 1000, 150.0)]]

ICUB_CONTACT = dict(kp=30.0, lxp=0.14, lxn=0.06, lyp=0.045, lyn=0.045, lz=0.065, fmin=5.0, fmax=1500.0, mu=0.3, normal="[0, 0, -1]")
ICUB = [
    ("lh", "se3", dict(tracked="l_hand", weight=1.0, kp=30.0, mask="111111")),
    ("rh", "se3", dict(tracked="r_hand", weight=1.0, kp=30.0, mask="111111")),
    ("lf", "se3", dict(tracked="left_foot", weight=1000.0, kp=30.0, mask="111111")),
    ("rf", "se3", dict(tracked="right_foot", weight=10.0, kp=30.0, mask="111111")),
    ("com", "com", dict(weight=3000.0, kp=50.0, mask="111")),
    ("momentum", "momentum", dict(weight=1000.0, kp=30.0, mask="000110")),
    ("posture", "posture", dict(weight=0.05, kp=10.0, ref="inria_start")),
    ("torso", "se3", dict(tracked="chest", weight=1.0, kp=30.0, mask="000111")),
    ("head", "se3", dict(tracked="head", weight=10.0, kp=30.0, mask="110111")),
    ("bounds", "bounds", dict(weight=1000.0)),
    ("contact_lfoot", "contact", dict(joint="l_ankle_roll", **ICUB_CONTACT)),
    ("contact_rfoot", "contact", dict(joint="r_ankle_roll", **ICUB_CONTACT)),
    ("self_collision-left", "self-collision", dict(tracked="l_wrist_yaw", radius=0.05, weight=500, kp=50.0, kd=250.0, margin=0.02, m=0.2)),
    ("self_collision-right", "self-collision", dict(tracked="r_wrist_yaw", radius=0.05, weight=500, kp=50.0, kd=250.0, margin=0.02, m=0.2)),
]
FRANKA = [("ee", "se3", dict(tracked="panda_joint7", weight=100.0, kp=30.0, mask="111111")),
          ("posture", "posture", dict(weight=0.75, kp=30.0, ref="start"))]
TIAGO = [("ee", "se3", dict(tracked="gripper_link", weight=1500.0, kp=30.0, mask="111111")),
         ("head", "se3", dict(tracked="head_2_link", weight=500.0, kp=30.0, mask="000111")),
         ("posture", "posture", dict(weight=0.1, kp=10.0, ref="start")),
         ("bounds", "bounds", dict(weight=10000))] + \
        [(n, "self-collision", dict(tracked=n, radius=0.1, weight=1000, kp=50.0, kd=250.0, margin=0.02, m=0.2))
         for n in ("sc-gripper", "sc-wrist", "sc-forearm", "sc-elbow")]

ROBOTS = {"talos": (TALOS, 50, 44, True), "icub": (ICUB, 38, 32, True), "franka": (FRANKA, 9, 9, False), "tiago": (TIAGO, 12, 12, False)}


def emit_tasks(tasks):
    out = []
    for name, typ, fields in tasks:
        out.append("%s:" % name)
        out.append("  type: %s" % typ)
        for k, v in fields.items():
            if isinstance(v, dict):  # nested map (self-collision `avoided:` frame -> radius)
                out.append("  %s:" % k)
                for kk, vv in v.items():
                    out.append("    %s: %s" % (kk, vv))
            else:
                out.append("  %s: %s" % (k, v))
    return "\n".join(out) + "\n"


def talos_with_avoided(tasks=None, stack=None):
    from inria_wbc_amd import model as mdl
    tasks = TALOS if tasks is None else tasks
    av = {n["name"]: n["avoided"] for n in (mdl.talos_stack() if stack is None else stack) if n["type"] == "self-collision"}
    out = []
    for name, typ, fields in tasks:
        if typ == "self-collision":
            f = dict(fields)
            kp, kd, margin, m = f.pop("kp"), f.pop("kd"), f.pop("margin"), f.pop("m")
            w = f.pop("weight")
            f["avoided"] = av[name]
            f.update(weight=w, kp=kp, kd=kd, margin=margin, m=m)
            fields = f
        out.append((name, typ, fields))
    return out


def emit_model_files():
    """The URDF's stand-in for the facade (robots/robot_wrapper.hpp), the virtual frames in the reference's own frames.yaml
    schema (etc/talos/frames.yaml) and a CONTROLLER tree that uses them."""
    from inria_wbc_amd import model as mdl
    d = os.path.join(ROOT, "configs", "talos")
    os.makedirs(d, exist_ok=True)
    m = mdl.talos_like()
    virtual = ["v_leg_right_3", "v_leg_left_3", "v_base_link_left", "v_base_link_right"]
    mdl.to_yaml(m, os.path.join(d, "talos_like.model.yaml"), skip_frames=virtual, ref_name="inria_start")
    with open(os.path.join(d, "frames.yaml"), "w") as f:
        f.write("# virtual frames (schema and values of inria_wbc's etc/talos/frames.yaml)\n")
        for name, ref, pos in (("v_leg_right_3", "leg_right_3_joint", [0.0, -0.1, -0.2]), ("v_leg_left_3", "leg_left_3_joint", [0, 0.1, -0.2]),
                               ("v_base_link_left", "base_link", [0.0, -0.1, 0]), ("v_base_link_right", "base_link", [0, 0.1, 0])):
            f.write("%s:\n  ref: \"%s\"\n  pos: %s\n" % (name, ref, pos))
    with open(os.path.join(d, "pos_tracker_model.yaml"), "w") as f:
        f.write("# CONTROLLER tree with the step before the path on the device: `model` stands where inria_wbc has `urdf`\n")
        f.write("# (a parsed tree, see robots/robot_wrapper.hpp), `frames` / `ref_config` / `tasks` as in talos_pos_tracker.yaml\n")
        f.write("CONTROLLER:\n  name: pos-tracker\n  solver: hip-batched\n  base_path: .\n  model: talos_like.model.yaml\n  frames: frames.yaml\n")
        f.write("  ref_config: inria_start\n  tasks: tasks.yaml\n  dt: 0.001\n  floating_base: true\n  closed_loop: false\n  verbose: false\n  batch: 8\n")
    with open(os.path.join(d, "walk_on_spot.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of the walk on the spot (schema and values of inria_wbc's etc/talos/walk_on_spot.yaml)\n")
        f.write("BEHAVIOR:\n  name: humanoid::walk-on-spot\n  traj_com_duration: 1\n  traj_foot_duration: 1\n  step_height: 0.1\n")
        f.write("  customize_task_weights:\n    momentum: 0.0\n")
    with open(os.path.join(d, "walk.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of the walk (schema of inria_wbc's etc/talos/walk.yaml; two cycles instead of ten)\n")
        f.write("BEHAVIOR:\n  name: humanoid::walk\n  traj_com_duration: 1\n  traj_foot_duration: 1\n  step_height: 0.1\n  step_length: 0.2\n")
        f.write("  num_of_cycles: 2\n  customize_task_weights:\n    momentum: 0.0\n")
    with open(os.path.join(d, "clapping.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of the clapping (schema of inria_wbc's etc/talos/clapping.yaml; motion_size sized for the Talos-like arms)\n")
        f.write("BEHAVIOR:\n  name: humanoid::clapping\n  trajectory_duration: 1.0\n  motion_size: 0.1\n")
    with open(os.path.join(d, "move_feet.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of the feet move (schema and values of inria_wbc's etc/talos/move_feet.yaml)\n")
        f.write("BEHAVIOR:\n  name: humanoid::move-feet\n  task_names: [lf, rf]\n  contact_names: [contact_lfoot, contact_rfoot]\n")
        f.write("  relative_targets_pos: [[0.05, 0.05, 0.05], [0.05, 0.05, 0.05]]\n  relative_targets_rpy: [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]\n")
        f.write("  trajectory_duration: 4\n  loop: false\n")


def emit_franka_model_files():
    """BASELINE config 1 on the model: Franka-like tree, CONTROLLER tree with it, the line behaviour of etc/franka/cartesian_line.yaml."""
    from inria_wbc_amd import model as mdl
    d = os.path.join(ROOT, "configs", "franka")
    os.makedirs(d, exist_ok=True)
    mdl.to_yaml(mdl.franka_like(), os.path.join(d, "franka_like.model.yaml"), ref_name="start")
    with open(os.path.join(d, "pos_tracker_model.yaml"), "w") as f:
        f.write("# CONTROLLER tree with the step before the path on the device (`model` stands where inria_wbc has `urdf`)\n")
        f.write("CONTROLLER:\n  name: pos-tracker\n  solver: hip-batched\n  base_path: .\n  model: franka_like.model.yaml\n")
        f.write("  ref_config: start\n  tasks: tasks.yaml\n  dt: 0.001\n  floating_base: false\n  closed_loop: false\n  verbose: false\n  batch: 4\n")
    with open(os.path.join(d, "cartesian_line.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of the line (schema and values of inria_wbc's etc/franka/cartesian_line.yaml)\n")
        f.write("BEHAVIOR:\n  name: generic::cartesian\n  task_names: [ee]\n  trajectory_duration: 2\n  relative_targets_pos: [[-0.4, 0, 0]]\n")
        f.write("  relative_targets_rpy: [[], []]\n  loop: true\n")


def emit_icub_model_files():
    """BASELINE config 3's robot on the model: iCub-like tree (z-down ankle frames), virtual frames of etc/icub/frames.yaml, a
    CONTROLLER tree and a small squat."""
    from inria_wbc_amd import model as mdl
    d = os.path.join(ROOT, "configs", "icub")
    os.makedirs(d, exist_ok=True)
    mdl.to_yaml(mdl.icub_like(), os.path.join(d, "icub_like.model.yaml"), skip_frames=["v_leg_right", "v_leg_left"], ref_name="inria_start")
    with open(os.path.join(d, "frames.yaml"), "w") as f:
        f.write("# virtual frames (schema and values of inria_wbc's etc/icub/frames.yaml)\n")
        f.write("v_leg_right:\n  ref: \"r_hip_yaw\"\n  pos: [0.0, -0.12, 0.0]\nv_leg_left:\n  ref: \"l_hip_yaw\"\n  pos: [0.0, -0.12, 0.0]\n")
    with open(os.path.join(d, "pos_tracker_model.yaml"), "w") as f:
        f.write("# CONTROLLER tree with the step before the path on the device (`model` stands where inria_wbc has `urdf`)\n")
        f.write("CONTROLLER:\n  name: pos-tracker\n  solver: hip-batched\n  base_path: .\n  model: icub_like.model.yaml\n  frames: frames.yaml\n")
        f.write("  ref_config: inria_start\n  tasks: tasks.yaml\n  dt: 0.001\n  floating_base: true\n  closed_loop: false\n  verbose: false\n  batch: 4\n")
    with open(os.path.join(d, "squat.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of a small squat (schema of inria_wbc's etc/talos/squat.yaml; 5 cm for the child-sized robot)\n")
        f.write("BEHAVIOR:\n  name: humanoid::move_com\n  trajectory_duration: 1\n  targets: [[0, 0, -0.05]]\n  mask: 001\n  absolute: false\n  loop: true\n")


def main():
    from inria_wbc_amd import model as mdl
    ROBOTS["talos"] = (talos_with_avoided(), 50, 44, True)
    ROBOTS["icub"] = (talos_with_avoided(ICUB, mdl.icub_stack()), 38, 32, True)
    emit_model_files()
    emit_franka_model_files()
    emit_icub_model_files()
    for robot, (tasks, nv, na, fb) in ROBOTS.items():
        d = os.path.join(ROOT, "configs", robot)
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "tasks.yaml"), "w") as f:
            f.write("# task stack of %s in inria_wbc's tasks.yaml schema (generated by tools/emit_configs.py)\n" % robot)
            f.write(emit_tasks(tasks))
        with open(os.path.join(d, "pos_tracker.yaml"), "w") as f:
            f.write("# CONTROLLER tree (schema of inria_wbc's <robot>_pos_tracker.yaml); nv / na replace the URDF lookup;\n")
            f.write("# base_path is filled in by the harness (relative paths are resolved against it)\n")
            f.write("CONTROLLER:\n  name: pos-tracker\n  solver: hip-batched\n  base_path: .\n  tasks: tasks.yaml\n")
            f.write("  dt: 0.001\n  floating_base: %s\n  closed_loop: false\n  verbose: false\n  nv: %d\n  na: %d\n" % ("true" if fb else "false", nv, na))
    with open(os.path.join(ROOT, "configs", "talos", "squat.yaml"), "w") as f:
        f.write("# BEHAVIOR tree of the squat (schema of inria_wbc's etc/talos/squat.yaml)\n")
        f.write("BEHAVIOR:\n  name: humanoid::move_com\n  trajectory_duration: 2\n  targets: [[0, 0, -0.2]]\n  mask: 001\n  absolute: false\n  loop: true\n")


if __name__ == "__main__":
    main()
