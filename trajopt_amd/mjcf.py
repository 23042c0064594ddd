"""MJCF front end: model XML text -> `ModelSpec` for the model compiler.

The reference passes its XML asset to the simulator (`trajopt/envs/mujoco_env.py:39-40`, asset
`trajopt/envs/assets/sawyer.xml`); this reads the subset of that format the compiler understands:

    <compiler angle="radian" coordinate="local" inertiafromgeom="true"/>
    <option timestep gravity integrator="Euler|RK4"/>
    <default> <joint .../> <geom .../> <motor .../> </default>      (top-level defaults only)
    <worldbody> nested <body pos> with <joint type="hinge|slide|ball|free" axis range damping armature limited>
                or <freejoint/> (free: position + unit quaternion in qpos, world-frame linear and body-frame
                angular velocity in qvel; ball: quaternion only),
                <geom type="sphere|capsule|plane" size pos fromto contype conaffinity margin>, <site name pos>
    <actuator> <motor joint gear ctrlrange/> </actuator>

Anything else that would change the dynamics (body orientations, joint offsets, other geom or joint
types, tendons, equality constraints, implicit integrators) raises `ValueError`; purely visual
elements and attributes are ignored.
"""
import xml.etree.ElementTree as ET

from .model import Body, Geom, Joint, ModelSpec

_VISUAL = {"light", "camera"}


def _floats(text, n=None):
    vals = tuple(float(x) for x in text.split())
    if n is not None and len(vals) != n:
        raise ValueError("expected %d numbers, got %r" % (n, text))
    return vals


def _merged(defaults, elem):
    d = dict(defaults)
    d.update(elem.attrib)
    return d


def load_mjcf(source, name, frame_skip, hand_site="finger") -> ModelSpec:
    """`source`: path of an XML file or the XML text itself."""
    text = source if source.lstrip().startswith("<") else open(source).read()
    root = ET.fromstring(text)
    if root.tag != "mujoco":
        raise ValueError("not a model file: root element %r" % root.tag)

    comp = root.find("compiler")
    if comp is not None:
        if comp.get("angle", "degree") != "radian":
            raise ValueError("only angle=\"radian\" is supported")
        if comp.get("coordinate", "local") != "local":
            raise ValueError("only coordinate=\"local\" is supported")
    else:
        raise ValueError("a <compiler angle=\"radian\"> element is required (the format's default is degrees)")
    opt = root.find("option")
    oa = {} if opt is None else opt.attrib
    integrator = oa.get("integrator", "Euler")
    if integrator not in ("Euler", "RK4"):
        raise ValueError("integrator %r has no kernel (Euler and RK4 only)" % integrator)
    timestep = float(oa.get("timestep", 0.002))
    gravity = _floats(oa.get("gravity", "0 0 -9.81"), 3)

    defaults = {"joint": {}, "geom": {}, "motor": {}}
    dflt = root.find("default")
    if dflt is not None:
        for e in dflt:
            if e.tag == "default":
                raise ValueError("nested default classes are not supported")
            if e.tag in defaults:
                defaults[e.tag].update(e.attrib)

    motors = {}
    act = root.find("actuator")
    if act is not None:
        for e in act:
            if e.tag != "motor":
                raise ValueError("actuator type %r has no kernel (motor only)" % e.tag)
            a = _merged(defaults["motor"], e)
            motors[a["joint"]] = (float(a.get("gear", "1").split()[0]), _floats(a.get("ctrlrange", "0 0"), 2),
                                  a.get("ctrllimited", "false") == "true")
    for tag in ("tendon", "equality"):
        if root.find(tag) is not None and len(root.find(tag)):
            raise ValueError("<%s> elements are not supported" % tag)

    plane_z = [None]
    used = set()

    def geom(e):
        a = _merged(defaults["geom"], e)
        kind = a.get("type", "sphere")
        collide = int(a.get("contype", 1)) != 0 and int(a.get("conaffinity", 1)) != 0
        margin = float(a.get("margin", 0.0))
        if kind == "plane":
            return Geom("plane", 0.0, pos=_floats(a.get("pos", "0 0 0"), 3), collide=collide, margin=margin)
        size = _floats(a["size"])[0]
        if kind == "sphere":
            return Geom("sphere", size, pos=_floats(a.get("pos", "0 0 0"), 3), collide=collide, margin=margin)
        if kind == "capsule":
            if "fromto" not in a:
                raise ValueError("capsules need fromto=")
            return Geom("capsule", size, fromto=_floats(a["fromto"], 6), collide=collide, margin=margin)
        raise ValueError("geom type %r has no inertia rule" % kind)

    def joint(e):
        a = _merged(defaults["joint"], e)
        kind = a.get("type", "hinge")
        if kind not in ("hinge", "slide", "ball", "free"):
            raise ValueError("joint type %r is not a joint type of the format" % kind)
        if kind in ("ball", "free"):
            if a.get("limited", "false") == "true":
                raise ValueError("limited %s joints have no kernel" % kind)
            if a.get("name") in motors:
                raise ValueError("motors on %s joints have no kernel" % kind)
            if kind == "free" and (float(a.get("damping", 0.0)) != 0.0 or float(a.get("armature", 0.0)) != 0.0):
                raise ValueError("free joints take no damping or armature")
            return Joint(axis=(0.0, 0.0, 1.0), range=(0.0, 0.0), damping=float(a.get("damping", 0.0)),
                         armature=float(a.get("armature", 0.0)), limited=False, gear=0.0, kind=kind)
        if any(abs(x) > 0 for x in _floats(a.get("pos", "0 0 0"), 3)):
            raise ValueError("joints must sit at the body origin")
        jn = a.get("name")
        gear, ctrl, ctrllimited = motors.get(jn, (0.0, (0.0, 0.0), True))
        if jn in motors:
            used.add(jn)
            if not ctrllimited:
                ctrl = (-float("inf"), float("inf"))
        limited = a.get("limited", "false") == "true"
        return Joint(axis=_floats(a.get("axis", "0 0 1"), 3), range=_floats(a.get("range", "0 0"), 2),
                     damping=float(a.get("damping", 0.0)), armature=float(a.get("armature", 0.0)), limited=limited,
                     gear=gear, ctrlrange=ctrl, kind=kind)

    def body(e):
        for att in ("quat", "euler", "axisangle", "xyaxes", "zaxis"):
            if att in e.attrib:
                raise ValueError("body orientation (%s=) is not supported" % att)
        joints, geoms, sites, children = [], [], {}, []
        for c in e:
            if c.tag == "joint":
                joints.append(joint(c))
            elif c.tag == "freejoint":
                joints.append(Joint(axis=(0.0, 0.0, 1.0), range=(0.0, 0.0), damping=0.0, armature=0.0, limited=False,
                                    gear=0.0, kind="free"))
            elif c.tag == "geom":
                geoms.append(geom(c))
            elif c.tag == "site":
                sites[c.get("name")] = _floats(c.get("pos", "0 0 0"), 3)
            elif c.tag == "body":
                children.append(body(c))
            elif c.tag == "inertial":
                raise ValueError("explicit <inertial> is not supported (inertia comes from the geoms)")
            elif c.tag not in _VISUAL:
                raise ValueError("unsupported element <%s> in a body" % c.tag)
        return Body(e.get("name", ""), _floats(e.get("pos", "0 0 0"), 3),
                    joints[0] if len(joints) == 1 else None, geoms, sites, children,
                    joints=joints if len(joints) > 1 else None)

    world = root.find("worldbody")
    if world is None:
        raise ValueError("no <worldbody>")
    roots = []
    for c in world:
        if c.tag == "body":
            roots.append(body(c))
        elif c.tag == "geom":
            g = geom(c)
            if g.kind != "plane":
                raise ValueError("only plane geoms may be attached to the world")
            if g.collide:
                plane_z[0] = g.pos[2]
        elif c.tag not in _VISUAL and c.tag != "site":
            raise ValueError("unsupported element <%s> in the world body" % c.tag)
    missing = set(motors) - used
    if missing:
        raise ValueError("motors refer to unknown joints: %s" % sorted(missing))

    kw = {}
    if opt is not None and "o_solref" in oa:
        kw["solref"] = _floats(oa["o_solref"], 2)
    return ModelSpec(name, timestep, frame_skip, roots, plane_z=plane_z[0], gravity=gravity, hand_site=hand_site,
                     integrator=integrator, **kw)
