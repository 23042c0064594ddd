"""Model compiler: kinematic-tree description -> flat arrays for the rollout kernels.

The reference hands its model file to MuJoCo (``trajopt/envs/mujoco_env.py:39-40``);
here the same information is compiled once on the host into a flat ``ModelDesc``
(links with hinge joints, welded bodies merged into their parent link, inertia
derived from the geoms, soft-constraint constants) that is (a) passed through
the C ABI (``mppi_model_desc`` in ``include/mppi_hip.h``) and (b) baked into
the HIP kernels as compile-time constants by ``tools/gen_model_header.py``.

Covered subset of the model language: bodies without orientation offsets, hinge
slide, ball and free joints at the body origin (several per body compose in order),
sphere / capsule / plane geoms, sites, motors with gear and ctrlrange, gravity, Euler
integrator.  A ball joint keeps its orientation as a unit quaternion in qpos (nq = nv + 1
per ball) and its velocity as the angular velocity in the body frame, like the free and
ball joints of the reference's simulator; a free joint is three world-aligned slides and
a ball.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

DENSITY = 1000.0  # default geom density
HINGE, SLIDE, BALL_X, BALL_Y, BALL_Z = 0, 1, 2, 3, 4     # a ball joint is three links in one frame


# --------------------------------------------------------------------------
# source-level description
# --------------------------------------------------------------------------
@dataclass
class Geom:
    kind: str                                  # 'sphere' | 'capsule' | 'plane'
    size: float = 0.0
    pos: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    fromto: Optional[Tuple[float, ...]] = None
    collide: bool = False                      # contype=1 conaffinity=1
    margin: float = 0.002


@dataclass
class Joint:
    axis: Tuple[float, float, float]
    range: Tuple[float, float]
    damping: float = 0.8
    armature: float = 0.004
    limited: bool = True
    gear: float = 10.0
    ctrlrange: Tuple[float, float] = (-1.0, 1.0)
    kind: str = "hinge"                         # 'hinge' | 'slide' | 'ball' (quaternion) | 'free' (3 slides + ball)


@dataclass
class Body:
    name: str
    pos: Tuple[float, float, float]
    joint: Optional[Joint] = None
    geoms: List[Geom] = field(default_factory=list)
    sites: dict = field(default_factory=dict)          # name -> local pos
    children: List["Body"] = field(default_factory=list)
    joints: Optional[List[Joint]] = None               # several joints on one body, applied in order

    def joint_list(self):
        if self.joints:
            return list(self.joints)
        return [] if self.joint is None else [self.joint]


@dataclass
class ModelSpec:
    name: str
    timestep: float
    frame_skip: int
    root: List[Body]                                    # children of the world body
    plane_z: Optional[float] = None                     # colliding world plane z = plane_z
    solref: Tuple[float, float] = (0.02, 1.0)
    solimp: Tuple[float, float, float, float, float] = (0.9, 0.95, 0.001, 0.5, 2.0)
    gravity: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    hand_site: str = "finger"
    integrator: str = "Euler"                           # 'Euler' (joint damping implicit) | 'RK4'


def reacher_7dof_spec() -> ModelSpec:
    """The 7-DoF arm of ``trajopt/envs/assets/sawyer.xml`` (line numbers below refer to it).

    option: timestep 0.01, gravity 0, Euler (:3); joint defaults armature 0.004,
    damping 0.8, limited (:5); geom default margin 0.002, no collision (:6);
    frame_skip = 2 (``trajopt/envs/reacher_env.py:29``).
    """
    cap = lambda r, *ft: Geom("capsule", r, fromto=tuple(ft))
    sph = lambda r, *p, **kw: Geom("sphere", r, pos=tuple(p), **kw)
    wrist_roll = Body("r_wrist_roll_link", (0, 0, 0),                                     # :47
                      Joint((1, 0, 0), (-1.5, 1.5)),                                       # :48, motor :108
                      [sph(0.08, 0.03, 0, 0, collide=True)],                               # :56
                      {"finger": (0.0, 0.0, 0.0)})                                         # :57
    wrist_flex = Body("r_wrist_flex_link", (0.321, 0, 0),                                 # :43
                      Joint((0, 1, 0), (-1.094, 0.0)),                                     # :45, motor :107
                      [cap(0.01, 0, -0.02, 0, 0, 0.02, 0)], children=[wrist_roll])         # :44
    forearm = Body("r_forearm_link", (0, 0, 0), None,                                     # :40 (welded)
                   [cap(0.05, 0, 0, 0, 0.291, 0, 0)], children=[wrist_flex])               # :41
    forearm_roll = Body("r_forearm_roll_link", (0, 0, 0),                                 # :36
                        Joint((1, 0, 0), (-1.5, 1.5)),                                     # :38, motor :106
                        [cap(0.02, -0.1, 0, 0, 0.1, 0, 0)], children=[forearm])            # :37
    elbow = Body("r_elbow_flex_link", (0.4, 0, 0),                                        # :32
                 Joint((0, 1, 0), (-2.3213, 0.0)),                                         # :34, motor :105
                 [cap(0.06, 0, -0.02, 0, 0, 0.02, 0)], children=[forearm_roll])            # :33
    upper_arm = Body("r_upper_arm_link", (0, 0, 0), None,                                 # :29 (welded)
                     [cap(0.06, 0, 0, 0, 0.4, 0, 0)], children=[elbow])                    # :30
    upper_roll = Body("r_upper_arm_roll_link", (0, 0, 0),                                 # :25
                      Joint((1, 0, 0), (-1.5, 1.7)),                                       # :27, motor :104
                      [cap(0.02, -0.1, 0, 0, 0.1, 0, 0)], children=[upper_arm])            # :26
    lift = Body("r_shoulder_lift_link", (0.1, 0, 0),                                      # :21
                Joint((0, 1, 0), (-0.5236, 1.3963), damping=2.0),                          # :23, motor :103
                [cap(0.1, 0, -0.1, 0, 0, 0.1, 0)], children=[upper_roll])                  # :22
    pan = Body("r_shoulder_pan_link", (0, -0.6, 0),                                       # :13
               Joint((0, 0, 1), (-2.2854, 1.714602), damping=2.0, gear=20.0),              # :19, motor :102
               [sph(0.05, -0.06, 0.05, 0.2), sph(0.05, 0.06, 0.05, 0.2),                   # :14-15
                sph(0.03, -0.06, 0.09, 0.2), sph(0.03, 0.06, 0.09, 0.2),                   # :16-17
                cap(0.1, 0, 0, -0.4, 0, 0, 0.2)], children=[lift])                         # :18
    return ModelSpec("reacher_7dof", 0.01, 2, [pan], plane_z=-0.425)                       # plane :11


def tree_18dof_spec() -> ModelSpec:
    """Synthetic 18-DoF tree for precision and scaling studies: a floating trunk (three slides
    along x, y, z, then three hinges about x, y, z) carrying four limbs of three hinges each.
    Gravity on, no joint limits, no collisions, frame_skip 1.  Not a model of the reference;
    the end-effector site `finger` is the tip of the first limb."""
    free = dict(limited=False, range=(0.0, 0.0))

    def limb(tag, root_pos, out):
        # out: unit direction of the limb in the trunk frame (+-x or +-y)
        o = np.asarray(out, float)
        side = (0.0, 1.0, 0.0) if abs(o[0]) > 0 else (1.0, 0.0, 0.0)
        seg = lambda L, r: Geom("capsule", r, fromto=(0, 0, 0) + tuple(L * o))
        tip = Body(tag + "_wrist", tuple(0.25 * o), Joint(tuple(abs(o)), damping=0.2, armature=0.01, gear=4.0, **free),
                   [seg(0.12, 0.025)], {"finger": tuple(0.12 * o)} if tag == "limb0" else {})
        mid = Body(tag + "_elbow", tuple(0.3 * o), Joint(side, damping=0.3, armature=0.01, gear=6.0, **free),
                   [seg(0.25, 0.03)], children=[tip])
        return Body(tag + "_shoulder", tuple(root_pos), Joint((0, 0, 1), damping=0.3, armature=0.01, gear=6.0, **free),
                    [seg(0.3, 0.035)], children=[mid])

    base_joints = [Joint(a, kind="slide", damping=0.0, armature=0.0, gear=0.0, **free) for a in ((1, 0, 0), (0, 1, 0), (0, 0, 1))]
    base_joints += [Joint(a, damping=0.0, armature=0.0, gear=0.0, **free) for a in ((1, 0, 0), (0, 1, 0), (0, 0, 1))]
    trunk = Body("trunk", (0.0, 0.0, 1.0), None,
                 [Geom("capsule", 0.09, fromto=(-0.2, 0, 0, 0.2, 0, 0)), Geom("sphere", 0.1, pos=(0.0, 0.0, 0.12))],
                 children=[limb("limb0", (0.22, 0.0, 0.0), (1, 0, 0)), limb("limb1", (-0.22, 0.0, 0.0), (-1, 0, 0)),
                           limb("limb2", (0.0, 0.12, 0.0), (0, 1, 0)), limb("limb3", (0.0, -0.12, 0.0), (0, -1, 0))],
                 joints=base_joints)
    return ModelSpec("tree_18dof", 0.005, 1, [trunk], gravity=(0.0, 0.0, -9.81))


def tree_18dof_quat_spec() -> ModelSpec:
    """`tree_18dof` with a true free joint on the trunk: position and unit quaternion in qpos (nq = 19),
    linear velocity in the world frame and angular velocity in the trunk frame in qvel (nv = 18)."""
    s = tree_18dof_spec()
    s.name = "tree_18dof_quat"
    trunk = s.root[0]
    trunk.joints = [Joint((0, 0, 1), kind="free", damping=0.0, armature=0.0, gear=0.0, limited=False, range=(0.0, 0.0))]
    return s


def tree_18dof_rk4_spec() -> ModelSpec:
    """`tree_18dof_quat` stepped with the fourth-order Runge-Kutta integrator (the integrator of the
    reference's `trajopt/envs/assets/hopper.xml:8`) at twice the time step."""
    s = tree_18dof_quat_spec()
    s.name = "tree_18dof_rk4"
    s.integrator = "RK4"
    s.timestep = 0.01
    return s


def strip_constraints(spec: ModelSpec) -> ModelSpec:
    """Copy of ``spec`` with joint limits and collisions removed (the 'no contacts' arm)."""
    import copy
    s = copy.deepcopy(spec)
    s.name = spec.name + "_free"
    s.plane_z = None

    def visit(b):
        for j in b.joint_list():
            j.limited = False
        for g in b.geoms:
            g.collide = False
        for c in b.children:
            visit(c)
    for b in s.root:
        visit(b)
    return s


# --------------------------------------------------------------------------
# inertia from geoms
# --------------------------------------------------------------------------
def _geom_inertia(g: Geom):
    """(mass, com, inertia about com) of one geom in its body frame, density 1000.

    The capsule rule is the one that reproduces the reference's stored MuJoCo
    transitions: a solid cylinder of radius r and length L + r centred on the
    segment midpoint (not the cylinder-plus-hemispheres formula of later
    MuJoCo releases).
    """
    if g.kind == "sphere":
        m = DENSITY * 4.0 / 3.0 * np.pi * g.size ** 3
        return m, np.asarray(g.pos, float), np.eye(3) * (0.4 * m * g.size ** 2)
    if g.kind == "capsule":
        a, b = np.asarray(g.fromto[:3], float), np.asarray(g.fromto[3:], float)
        L = np.linalg.norm(b - a)
        u = (b - a) / L
        r = g.size
        ell = L + r
        m = DENSITY * np.pi * r * r * ell
        i_ax = 0.5 * m * r * r
        i_pp = m * (3.0 * r * r + ell * ell) / 12.0
        I = i_pp * np.eye(3) + (i_ax - i_pp) * np.outer(u, u)
        return m, 0.5 * (a + b), I
    raise ValueError("no inertia rule for geom kind %r" % g.kind)


def _shift(I, m, d):
    """parallel-axis: inertia about a point displaced by d from the com"""
    return I + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))


# --------------------------------------------------------------------------
# flat model
# --------------------------------------------------------------------------
@dataclass
class ModelDesc:
    name: str
    nv: int
    h: float
    frame_skip: int
    parent: np.ndarray        # (nv,) int32, -1 = world
    pos: np.ndarray           # (nv,3) link origin in parent link frame
    axis: np.ndarray          # (nv,3) hinge axis, link frame
    mass: np.ndarray          # (nv,)
    com: np.ndarray           # (nv,3) link frame
    inertia: np.ndarray       # (nv,6) about com, link frame: xx yy zz xy xz yz
    armature: np.ndarray
    damping: np.ndarray
    limited: np.ndarray       # (nv,) int32
    range_lo: np.ndarray
    range_hi: np.ndarray
    gear: np.ndarray
    ctrl_lo: np.ndarray
    ctrl_hi: np.ndarray
    dof_invweight: np.ndarray  # (nv,) diag of inverse mass matrix at the reset pose
    # single sphere-vs-horizontal-plane contact (link < 0: none)
    contact_link: int
    contact_pos: np.ndarray   # (3,) sphere centre in link frame
    contact_radius: float
    contact_plane_z: float
    contact_margin: float
    contact_invweight: float
    # end-effector site
    hand_link: int
    hand_pos: np.ndarray      # (3,)
    # soft-constraint constants
    solref_k: float
    solref_b: float
    solimp: np.ndarray        # (5,)
    jtype: np.ndarray = None  # (nv,) int32: 0 hinge, 1 slide, 2/3/4 x/y/z component of a ball joint
    gravity: np.ndarray = None  # (3,)
    nq: int = None            # length of qpos: nv + number of ball joints
    qadr: np.ndarray = None   # (nv,) int32: qpos index of the joint's coordinate (a ball's three links: its quaternion w)
    integrator: int = 0       # 0 semi-implicit Euler with implicit joint damping, 1 fourth-order Runge-Kutta

    def __post_init__(self):
        if self.jtype is None:
            self.jtype = np.zeros(self.nv, np.int32)
        if self.gravity is None:
            self.gravity = np.zeros(3)
        if self.qadr is None:
            adr, n = [], 0
            for t in self.jtype:
                if t == BALL_X:
                    adr.append(n); n += 4
                elif t in (BALL_Y, BALL_Z):
                    adr.append(adr[-1])
                else:
                    adr.append(n); n += 1
            self.qadr, self.nq = np.asarray(adr, np.int32), n

    def qpos0(self):
        """Reference configuration: zeros, unit quaternions."""
        q = np.zeros(self.nq)
        q[self.qadr[self.jtype == BALL_X]] = 1.0
        return q

    def link_inertia_matrix(self, i):
        xx, yy, zz, xy, xz, yz = self.inertia[i]
        return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def _rot(axis, ang):
    a = np.asarray(axis, float)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


def _quat_rot(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def mass_matrix(md: ModelDesc, q):
    """Joint-space inertia (with armature) by the composite-body method; host-side
    helper used by the model compiler for the constraint weights."""
    nv = md.nv
    R = [None] * nv; p = [None] * nv
    for i in range(nv):
        pi = md.parent[i]
        Rp = np.eye(3) if pi < 0 else R[pi]
        pp = np.zeros(3) if pi < 0 else p[pi]
        a = md.qadr[i]
        if md.jtype[i] == SLIDE:
            p[i] = pp + Rp @ (md.pos[i] + q[a] * md.axis[i])
            R[i] = Rp
        elif md.jtype[i] == BALL_X:
            p[i] = pp + Rp @ md.pos[i]
            R[i] = Rp @ _quat_rot(*q[a:a + 4])
        elif md.jtype[i] in (BALL_Y, BALL_Z):
            p[i], R[i] = pp, Rp                      # same frame as the ball's first link
        else:
            p[i] = pp + Rp @ md.pos[i]
            R[i] = Rp @ _rot(md.axis[i], q[a])
    # spatial inertia about the world origin: (m, m*c, I_O)
    m = md.mass.copy(); hc = np.zeros((nv, 3)); IO = np.zeros((nv, 3, 3))
    S = np.zeros((nv, 6))
    for i in range(nv):
        c = p[i] + R[i] @ md.com[i]
        Ic = R[i] @ md.link_inertia_matrix(i) @ R[i].T
        hc[i] = m[i] * c
        IO[i] = _shift(Ic, m[i], c)
        a = R[i] @ md.axis[i]
        if md.jtype[i] == SLIDE:
            S[i, 3:] = a
        else:
            S[i, :3] = a; S[i, 3:] = np.cross(p[i], a)
    for i in range(nv - 1, -1, -1):
        pi = md.parent[i]
        if pi >= 0:
            m[pi] += m[i]; hc[pi] += hc[i]; IO[pi] += IO[i]
    M = np.zeros((nv, nv))
    for j in range(nv):
        w, v = S[j, :3], S[j, 3:]
        lin = m[j] * v + np.cross(w, hc[j])
        ang = IO[j] @ w + np.cross(hc[j], v)
        i = j
        while i >= 0:
            M[i, j] = M[j, i] = S[i, :3] @ ang + S[i, 3:] @ lin
            i = md.parent[i]
    return M + np.diag(md.armature), (R, p)


def compile_model(spec: ModelSpec) -> ModelDesc:
    links = []      # dict per link

    def add_geoms(link, body, offset):
        for g in body.geoms:
            if g.kind == "plane":
                continue
            m, c, I = _geom_inertia(g)
            link["parts"].append((m, c + offset, I))
            if g.collide:
                link["contact"] = (np.asarray(g.pos, float) + offset, g.size, g.margin,
                                   body.name)
        for name, sp in body.sites.items():
            link["sites"][name] = np.asarray(sp, float) + offset
        link.setdefault("bodies", []).append((body.name, offset.copy()))

    def visit(body, parent_link, offset_in_parent):
        pos = np.asarray(body.pos, float)
        joints = body.joint_list()
        if not joints:
            if parent_link < 0:
                raise ValueError("static bodies attached to the world are not supported")
            off = offset_in_parent + pos
            add_geoms(links[parent_link], body, off)
            for ch in body.children:
                visit(ch, parent_link, off)
            return
        # one link per degree of freedom; the body's geoms ride on the last one, the others are massless
        import dataclasses
        prim = []
        for jn in joints:
            unit = lambda k, kind: dataclasses.replace(jn, axis=tuple(float(c == k) for c in range(3)), kind=kind)
            if jn.kind == "free":
                prim += [unit(k, "slide") for k in range(3)]
            if jn.kind in ("free", "ball"):
                if jn.limited:
                    raise ValueError("limited ball joints are not supported")
                prim += [unit(0, "ball_x"), unit(1, "ball_y"), unit(2, "ball_z")]
            elif jn.kind in ("hinge", "slide"):
                prim.append(jn)
            else:
                raise ValueError("joint kind %r" % (jn.kind,))
        joints = prim
        for n, jn in enumerate(joints):
            link = dict(parent=parent_link, pos=(offset_in_parent + pos) if n == 0 else np.zeros(3), joint=jn,
                        parts=[], sites={}, contact=None)
            links.append(link)
            parent_link = len(links) - 1
        idx = len(links) - 1
        add_geoms(link, body, np.zeros(3))
        for ch in body.children:
            visit(ch, idx, np.zeros(3))

    for b in spec.root:
        visit(b, -1, np.zeros(3))

    nv = len(links)
    f = lambda: np.zeros(nv)
    md = ModelDesc(
        name=spec.name, nv=nv, h=spec.timestep, frame_skip=spec.frame_skip,
        parent=np.array([l["parent"] for l in links], np.int32),
        pos=np.array([l["pos"] for l in links]), axis=np.zeros((nv, 3)),
        mass=f(), com=np.zeros((nv, 3)), inertia=np.zeros((nv, 6)),
        armature=f(), damping=f(), limited=np.zeros(nv, np.int32), range_lo=f(), range_hi=f(),
        gear=f(), ctrl_lo=f(), ctrl_hi=f(), dof_invweight=f(),
        contact_link=-1, contact_pos=np.zeros(3), contact_radius=0.0,
        contact_plane_z=0.0, contact_margin=0.0, contact_invweight=0.0,
        hand_link=-1, hand_pos=np.zeros(3), solref_k=0.0, solref_b=0.0,
        solimp=np.asarray(spec.solimp, float),
        jtype=np.array([dict(hinge=HINGE, slide=SLIDE, ball_x=BALL_X, ball_y=BALL_Y, ball_z=BALL_Z)[l["joint"].kind]
                        for l in links], np.int32),
        gravity=np.asarray(spec.gravity, float),
        integrator={"Euler": 0, "RK4": 1}[spec.integrator])
    contact_body = None
    for i, l in enumerate(links):
        j = l["joint"]
        ax = np.asarray(j.axis, float)
        md.axis[i] = ax / np.linalg.norm(ax)
        md.armature[i], md.damping[i] = j.armature, j.damping
        md.limited[i] = int(j.limited)
        md.range_lo[i], md.range_hi[i] = j.range
        md.gear[i] = j.gear
        md.ctrl_lo[i], md.ctrl_hi[i] = j.ctrlrange
        if l["parts"]:
            m = sum(p[0] for p in l["parts"])
            c = sum(p[0] * p[1] for p in l["parts"]) / m
            I = sum(_shift(p[2], p[0], c - p[1]) for p in l["parts"])
            md.mass[i], md.com[i] = m, c
            md.inertia[i] = [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]]
        if spec.hand_site in l["sites"]:
            md.hand_link, md.hand_pos = i, l["sites"][spec.hand_site]
        if l["contact"] is not None and spec.plane_z is not None:
            md.contact_link = i
            md.contact_pos, md.contact_radius, md.contact_margin, contact_body = l["contact"]
            md.contact_plane_z = spec.plane_z

    # soft-constraint constants (solref = time constant / damping ratio form)
    d0, dmax, width, mid, power = spec.solimp
    tc = max(spec.solref[0], 2.0 * spec.timestep)
    dr = spec.solref[1]
    md.solref_k = 1.0 / (dmax * dmax * tc * tc * dr * dr)
    md.solref_b = 2.0 / (dmax * tc)

    # constraint weights at the reset pose q = 0
    M0, (R, p) = mass_matrix(md, md.qpos0())
    Minv = np.linalg.inv(M0)
    md.dof_invweight = np.diag(Minv).copy()
    if md.contact_link >= 0:
        # translational inverse weight of the colliding body: Jacobian of that
        # body's own centre of mass (welded bodies keep their own com)
        l = links[md.contact_link]
        body_parts = None
        # recompute the colliding source body's com from its own geoms
        def find(body, off):
            nonlocal body_parts
            if body.name == contact_body:
                parts = [_geom_inertia(g) for g in body.geoms if g.kind != "plane"]
                mm = sum(x[0] for x in parts)
                body_parts = sum(x[0] * x[1] for x in parts) / mm + off
            for ch in body.children:
                find(ch, (off + np.asarray(ch.pos, float)) if not ch.joint_list() else np.zeros(3))
        for b in spec.root:
            find(b, np.zeros(3))
        cw = p[md.contact_link] + R[md.contact_link] @ body_parts
        J = np.zeros((3, nv))
        i = md.contact_link
        while i >= 0:
            a = R[i] @ md.axis[i]
            J[:, i] = np.cross(a, cw - p[i])
            i = md.parent[i]
        md.contact_invweight = float(np.trace(J @ Minv @ J.T) / 3.0)
    return md


_CACHE = {}


def get_model(name: str) -> ModelDesc:
    if name not in _CACHE:
        if name == "reacher_7dof":
            _CACHE[name] = compile_model(reacher_7dof_spec())
        elif name == "reacher_7dof_free":
            _CACHE[name] = compile_model(strip_constraints(reacher_7dof_spec()))
        elif name == "tree_18dof":
            _CACHE[name] = compile_model(tree_18dof_spec())
        elif name == "tree_18dof_quat":
            _CACHE[name] = compile_model(tree_18dof_quat_spec())
        elif name == "tree_18dof_rk4":
            _CACHE[name] = compile_model(tree_18dof_rk4_spec())
        else:
            raise KeyError(name)
    return _CACHE[name]
