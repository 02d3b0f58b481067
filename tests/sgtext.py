"""Writes scene graphs in the .sg text form (what `rth_load_sg` and the oracle read) from Python, for tests
that need scenes the YAML schema cannot express: CSG objects, triangles, gradient / ring / test patterns."""
import numpy as np


def translation(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def scaling(x, y, z):
    return np.diag([x, y, z, 1.0])


def rotation_x(r):
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = np.cos(r), -np.sin(r), np.sin(r), np.cos(r)
    return m


def rotation_y(r):
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = np.cos(r), np.sin(r), -np.sin(r), np.cos(r)
    return m


def rotation_z(r):
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = np.cos(r), -np.sin(r), np.sin(r), np.cos(r)
    return m


def view_transform(frm, to, up):
    """src/scene.rs:157-169."""
    frm, to, up = (np.asarray(v, dtype=float) for v in (frm, to, up))
    fwd = (to - frm) / np.linalg.norm(to - frm)
    left = np.cross(fwd, up / np.linalg.norm(up))
    true_up = np.cross(left, fwd)
    o = np.eye(4)
    o[0, :3], o[1, :3], o[2, :3] = left, true_up, -fwd
    return o @ translation(*(-frm))


class SG:
    def __init__(self):
        self.xfs, self.mats, self.defs, self.nodes, self.lights = [np.eye(4)], [], [], [], []
        self.camera = None
        self.top = 0
        self._open = []  # children still expected by the groups / csg nodes being written

    def xf(self, m):
        m = np.asarray(m, dtype=float)
        for i, q in enumerate(self.xfs):
            if np.array_equal(q, m):
                return i
        self.xfs.append(m)
        self.defs.append("xf %d %s" % (len(self.xfs) - 1, " ".join(repr(float(v)) for v in m.ravel())))
        return len(self.xfs) - 1

    def mat(self, color=(1, 1, 1), ambient=0.1, diffuse=0.9, specular=0.9, shininess=200.0, reflective=0.0, transparency=0.0,
            ri=1.0, pattern=None):
        """pattern: (kind, colour a, colour b, 4x4 transform) with kind in stripe / gradient / ring / checkers / test."""
        rec = "%s %s %s %r %r %r %r %r %r %r " % (*[repr(float(c)) for c in color], float(ambient), float(diffuse), float(specular),
                                                   float(shininess), float(reflective), float(transparency), float(ri))
        if pattern is None:
            rec += "none"
        else:
            kind, a, b, m = pattern
            pid = self.xf(m)
            rec += "%s %s %s %d" % (kind, " ".join(repr(float(c)) for c in a), " ".join(repr(float(c)) for c in b), pid)
        self.defs.append("mat %d %s" % (len(self.mats), rec))
        self.mats.append(rec)
        return len(self.mats) - 1

    def _node(self, text, children=0):
        if not self._open:
            self.top += 1
        else:
            self._open[-1] -= 1
        self.nodes.append("node " + text)
        while self._open and self._open[-1] == 0:
            self._open.pop()
        if children:
            self._open.append(children)

    def _head(self, kind, mat, xf, shadow):
        return "%s mat %d xf %d shadow %d" % (kind, mat, self.xf(xf) if not isinstance(xf, int) else xf, 1 if shadow else 0)

    def shape(self, kind, mat, xf=0, shadow=True, mn=None, mx=None, closed=False):
        t = self._head(kind, mat, xf, shadow)
        if kind in ("cylinder", "cone"):
            t += " %r %r %d" % (float(-np.inf if mn is None else mn), float(np.inf if mx is None else mx), 1 if closed else 0)
        self._node(t)

    def tri(self, mat, p1, p2, p3, normals=None, xf=0, shadow=True):
        t = self._head("stri" if normals is not None else "tri", mat, xf, shadow)
        t += " " + " ".join(repr(float(v)) for p in (p1, p2, p3) for v in p)
        if normals is not None:
            t += " " + " ".join(repr(float(v)) for n in normals for v in (*n, 0.0))
        self._node(t)

    def group(self, n_children, mat, xf=0, shadow=True):
        self._node(self._head("group", mat, xf, shadow) + " %d" % n_children, children=n_children)

    def csg(self, op, mat, xf=0, shadow=True):
        self._node(self._head("csg", mat, xf, shadow) + " " + op, children=2)

    def set_camera(self, w, h, fov, frm, to, up=(0, 1, 0)):
        self.camera = (w, h, fov, self.xf(view_transform(frm, to, up)))

    def light(self, at, intensity=(1, 1, 1)):
        self.lights.append("light %s %s" % (" ".join(repr(float(v)) for v in at), " ".join(repr(float(v)) for v in intensity)))

    def text(self):
        assert not self._open, "a group or csg node is missing children"
        cam = "camera %d %d %r %d" % self.camera
        return "\n".join(["rtsg 1"] + self.defs + [cam] + self.lights + ["objects %d" % self.top] + self.nodes + ["end"]) + "\n"

    def write(self, path):
        with open(path, "w") as f:
            f.write(self.text())
        return str(path)


def icosahedron():
    """12 vertices, 20 faces (outward winding), unit circumradius."""
    g = (1 + 5 ** 0.5) / 2
    v = np.array([(-1, g, 0), (1, g, 0), (-1, -g, 0), (1, -g, 0), (0, -1, g), (0, 1, g), (0, -1, -g), (0, 1, -g),
                  (g, 0, -1), (g, 0, 1), (-g, 0, -1), (-g, 0, 1)], dtype=float)
    v /= np.linalg.norm(v[0])
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    return v, f


def tetrahedron():
    v = np.array([(1, 1, 1), (1, -1, -1), (-1, 1, -1), (-1, -1, 1)], dtype=float) / 3 ** 0.5
    f = [(0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)]
    return v, f


def replay(path, rt):
    """Reads a .sg file and makes the calls of include/rtamd_host.h's scene builder that a host walking the same object
    tree would make (one per node, groups and CSG nodes bracketed by begin / end) -> rt.WorldBuilder."""
    xfs, mats = {0: np.eye(4)}, {}
    b = rt.WorldBuilder()
    lines = []
    for raw in open(path):
        t = raw.split("#")[0].split()
        if t:
            lines.append(t)
    assert lines[0][0] == "rtsg"
    pos = [1]

    def material(tok):
        m = rt.default_material()
        v = [float(x) for x in tok[:10]]
        m.color[:] = v[0:3]
        m.ambient, m.diffuse, m.specular, m.shininess, m.reflective, m.transparency, m.refractive_index = v[3:10]
        if tok[10] != "none":
            m.pattern = rt.PATTERNS[tok[10]]
            m.pat_a[:] = [float(x) for x in tok[11:14]]
            m.pat_b[:] = [float(x) for x in tok[14:17]]
            m.pat_transform[:] = list(xfs[int(tok[17])].ravel())
        return m

    def node():
        t = lines[pos[0]]
        pos[0] += 1
        assert t[0] == "node" and t[2] == "mat" and t[4] == "xf" and t[6] == "shadow", t
        kind, mat, xf, shadow, rest = t[1], mats[int(t[3])], xfs[int(t[5])], t[7] != "0", t[8:]
        if kind in ("sphere", "plane", "cube"):
            b.add(kind, xf, mat, shadow)
        elif kind in ("cylinder", "cone"):
            b.add(kind, xf, mat, shadow, mn=float(rest[0]), mx=float(rest[1]), closed=rest[2] != "0")
        elif kind == "tri":
            p = np.array([float(x) for x in rest[:9]]).reshape(3, 3)
            b.add("triangle", xf, mat, shadow, points=np.hstack([p, np.ones((3, 1))]))
        elif kind == "stri":
            p = np.array([float(x) for x in rest[:9]]).reshape(3, 3)
            n = np.array([float(x) for x in rest[9:21]]).reshape(3, 4)
            b.add("smooth_triangle", xf, mat, shadow, points=np.hstack([p, np.ones((3, 1))]), normals=n)
        elif kind == "group":
            b.group_begin(xf, mat, shadow)
            for _ in range(int(rest[0])):
                node()
            b.group_end()
        elif kind == "csg":
            b.csg_begin(rest[0], xf, mat, shadow)
            node()
            node()
            b.csg_end()
        else:
            raise AssertionError(kind)

    while pos[0] < len(lines):
        t = lines[pos[0]]
        pos[0] += 1
        if t[0] == "xf":
            xfs[int(t[1])] = np.array([float(x) for x in t[2:18]]).reshape(4, 4)
        elif t[0] == "mat":
            mats[int(t[1])] = material(t[2:])
        elif t[0] == "camera":
            b.camera(int(t[1]), int(t[2]), float(t[3]), xfs[int(t[4])])
        elif t[0] == "light":
            b.light([float(x) for x in t[1:4]], [float(x) for x in t[4:7]])
        elif t[0] == "objects":
            for _ in range(int(t[1])):
                node()
        elif t[0] == "end":
            break
        else:
            raise AssertionError(t[0])
    return b
