"""A small SPIR-V interpreter for fragment shaders, used to run the reference's own compiled shaders
(compiled_shaders/*.frag.spv) on the CPU and record golden input/output vectors for the oracle.

It executes one fragment invocation at a time with IEEE binary32 arithmetic (numpy float32), covers
exactly the instructions glslang emitted for the reference's hot-path shaders, and delegates texture
sampling to Python objects that follow the Vulkan image-operation rules for the sampler states the
reference creates (VulkanHelper.cs:123-220).  It is test tooling: it reads shader binaries from the
reference checkout at generation time only; nothing at test time or in the product needs it.
"""
from __future__ import annotations

import math
import struct

import numpy as np

F = np.float32


class Kill(Exception):
    pass


# --------------------------------------------------------------------------------------------------
# textures

class Tex2D:
    """2D image [H, W, C] float32 with a sampler: filter in {linear, nearest}, wrap in {repeat, clamp, border}."""

    def __init__(self, data, filt="linear", wrap="repeat", border=1.0):
        self.data = np.asarray(data, dtype=np.float32)
        self.filt, self.wrap, self.border = filt, wrap, F(border)
        self.h, self.w = self.data.shape[:2]

    def size(self):
        return (self.w, self.h)

    def _texel(self, i, j):
        c = self.data.shape[2]
        if self.wrap == "repeat":
            i, j = i % self.w, j % self.h
        elif self.wrap == "clamp":
            i, j = min(max(i, 0), self.w - 1), min(max(j, 0), self.h - 1)
        elif not (0 <= i < self.w and 0 <= j < self.h):
            return np.full(4, self.border, dtype=np.float32)
        t = self.data[j, i]
        out = np.array([0, 0, 0, 1], dtype=np.float32)
        out[:c] = t
        return out

    def sample(self, uv):
        u, v = F(uv[0]) * F(self.w), F(uv[1]) * F(self.h)
        if self.filt == "nearest":
            return self._texel(int(math.floor(u)), int(math.floor(v)))
        x, y = u - F(0.5), v - F(0.5)
        i0, j0 = int(math.floor(x)), int(math.floor(y))
        a, b = F(x - F(i0)), F(y - F(j0))
        t00, t10, t01, t11 = self._texel(i0, j0), self._texel(i0 + 1, j0), self._texel(i0, j0 + 1), self._texel(i0 + 1, j0 + 1)
        return ((F(1) - a) * (F(1) - b)) * t00 + (a * (F(1) - b)) * t10 + ((F(1) - a) * b) * t01 + (a * b) * t11


_FACE_AXES = {  # face -> (sc, tc) as (sign, axis) pairs; Vulkan cube map face selection table
    0: ((-1, 2), (-1, 1)), 1: ((+1, 2), (-1, 1)), 2: ((+1, 0), (+1, 2)),
    3: ((+1, 0), (-1, 2)), 4: ((+1, 0), (-1, 1)), 5: ((-1, 0), (-1, 1))}


def cube_face_st(d):
    ax = [abs(F(d[0])), abs(F(d[1])), abs(F(d[2]))]
    if ax[2] >= ax[0] and ax[2] >= ax[1]:
        face, ma = (4 if d[2] >= 0 else 5), ax[2]
    elif ax[1] >= ax[0]:
        face, ma = (2 if d[1] >= 0 else 3), ax[1]
    else:
        face, ma = (0 if d[0] >= 0 else 1), ax[0]
    (ss, sa), (ts, ta) = _FACE_AXES[face]
    sc, tc = F(ss) * F(d[sa]), F(ts) * F(d[ta])
    return face, F(0.5) * (sc / ma) + F(0.5), F(0.5) * (tc / ma) + F(0.5)


def face_point(face, s, t):
    """Point on the unit cube for face coordinates s, t in [-1, 1] (inverse of the table above)."""
    return {0: (1, -t, -s), 1: (-1, -t, s), 2: (s, 1, t), 3: (s, -1, -t), 4: (s, -t, 1), 5: (-s, -t, -1)}[face]


class TexCube:
    """Cube image [6, N, N, C] float32; filter linear (seamless) or nearest."""

    def __init__(self, data, filt="linear"):
        self.data = np.asarray(data, dtype=np.float32)
        self.filt = filt
        self.n = self.data.shape[1]

    def _texel_seamless(self, face, i, j):
        """Texel (i, j) of `face`, where one coordinate may be one step outside: the texel across the edge."""
        n = self.n
        out_i, out_j = not (0 <= i < n), not (0 <= j < n)
        if out_i and out_j:
            return None
        if out_i or out_j:
            # walk around the edge in 3D: take the centre of the outside texel on the face plane, push it
            # onto the neighbouring face by reflecting the overshoot around the shared edge
            s, t = (2 * i + 1) / n - 1, (2 * j + 1) / n - 1
            over = 1.0 / n
            s_c, t_c = max(-1.0, min(1.0, s)), max(-1.0, min(1.0, t))
            p = np.array(face_point(face, s_c, t_c), dtype=np.float64)
            major = int(np.argmax(np.abs(np.array(face_point(face, 0.0, 0.0)))))
            p[major] *= (1.0 - over)      # half a texel inside the neighbouring face
            f2, s2, t2 = cube_face_st(p)
            i, j = int(math.floor(float(s2) * n)), int(math.floor(float(t2) * n))
            face = f2
            i, j = min(max(i, 0), n - 1), min(max(j, 0), n - 1)
        c = self.data.shape[3]
        out = np.array([0, 0, 0, 1], dtype=np.float32)
        out[:c] = self.data[face, j, i]
        return out

    def sample(self, d):
        face, s, t = cube_face_st(d)
        n = self.n
        if self.filt == "nearest":
            i = min(max(int(math.floor(s * F(n))), 0), n - 1)
            j = min(max(int(math.floor(t * F(n))), 0), n - 1)
            return self._texel_seamless(face, i, j)
        x, y = s * F(n) - F(0.5), t * F(n) - F(0.5)
        i0, j0 = int(math.floor(x)), int(math.floor(y))
        a, b = F(x - F(i0)), F(y - F(j0))
        taps = [self._texel_seamless(face, i0 + di, j0 + dj) for dj in (0, 1) for di in (0, 1)]
        present = [t for t in taps if t is not None]
        if len(present) == 3:
            mean = (present[0] + present[1] + present[2]) / F(3)
            taps = [t if t is not None else mean for t in taps]
        w = [(F(1) - a) * (F(1) - b), a * (F(1) - b), (F(1) - a) * b, a * b]
        return w[0] * taps[0] + w[1] * taps[1] + w[2] * taps[2] + w[3] * taps[3]


# --------------------------------------------------------------------------------------------------
# module parsing

class Module:
    def __init__(self, path):
        data = open(path, "rb").read()
        w = struct.unpack("<%dI" % (len(data) // 4), data)
        if w[0] != 0x07230203:
            raise ValueError("not a SPIR-V module")
        self.types, self.consts, self.names, self.decor, self.member_decor = {}, {}, {}, {}, {}
        self.globals, self.functions, self.ext_sets = {}, {}, {}
        self.entry = None
        i, cur = 5, None
        while i < len(w):
            wc, op = w[i] >> 16, w[i] & 0xFFFF
            a = w[i + 1:i + wc]
            i += wc
            if op == 5:
                self.names[a[0]] = self._str(a[1:])
            elif op == 11:
                self.ext_sets[a[0]] = self._str(a[1:])
            elif op == 15:
                self.entry = a[1]
            elif op == 71:
                self.decor.setdefault(a[0], {})[a[1]] = a[2:] or (1,)
            elif op == 72:
                self.member_decor.setdefault((a[0], a[1]), {})[a[2]] = a[3:] or (1,)
            elif op == 19:
                self.types[a[0]] = ("void",)
            elif op == 20:
                self.types[a[0]] = ("bool",)
            elif op == 21:
                self.types[a[0]] = ("int", a[1], a[2])
            elif op == 22:
                self.types[a[0]] = ("float", a[1])
            elif op == 23:
                self.types[a[0]] = ("vec", a[1], a[2])
            elif op == 24:
                self.types[a[0]] = ("mat", a[1], a[2])
            elif op == 25:
                self.types[a[0]] = ("image", a[2])   # dim: 1 = 2D, 3 = Cube
            elif op == 27:
                self.types[a[0]] = ("sampled_image", a[1])
            elif op == 28:
                self.types[a[0]] = ("array", a[1], a[2])
            elif op == 30:
                self.types[a[0]] = ("struct", tuple(a[1:]))
            elif op == 32:
                self.types[a[0]] = ("ptr", a[1], a[2])
            elif op == 33:
                self.types[a[0]] = ("fn",)
            elif op == 43:
                self.consts[a[1]] = self._scalar_const(a[0], a[2])
            elif op == 44:
                self.consts[a[1]] = ("composite", a[0], tuple(a[2:]))
            elif op == 41 or op == 42:
                self.consts[a[1]] = bool(op == 41)
            elif op == 59 and cur is None:
                self.globals[a[1]] = {"type": a[0], "storage": a[2], "init": a[3] if len(a) > 3 else None}
            elif op == 54:
                cur = {"id": a[1], "params": [], "blocks": {}, "order": [], "result_type": a[0]}
                self.functions[a[1]] = cur
                block = None
            elif op == 55:
                cur["params"].append(a[1])
            elif op == 56:
                cur = None
            elif cur is not None:
                if op == 248:
                    block = []
                    cur["blocks"][a[0]] = block
                    cur["order"].append(a[0])
                else:
                    block.append((op, a))

    @staticmethod
    def _str(ws):
        b = b"".join(struct.pack("<I", x) for x in ws)
        return b.split(b"\0")[0].decode()

    def _scalar_const(self, ty, word):
        t = self.types[ty]
        if t[0] == "float":
            return F(struct.unpack("<f", struct.pack("<I", word))[0])
        if t[0] == "int":
            return np.int32(struct.unpack("<i", struct.pack("<I", word))[0]) if t[2] else np.uint32(word)
        raise ValueError(t)

    def const_value(self, cid):
        c = self.consts[cid]
        if isinstance(c, tuple) and c[0] == "composite":
            parts = [self.const_value(x) for x in c[2]]
            t = self.types[c[1]]
            if t[0] == "vec":
                return np.array(parts, dtype=np.float32 if self.types[t[1]][0] == "float" else np.int64)
            return parts  # arrays / matrices: list of element values
        return c


# --------------------------------------------------------------------------------------------------
# execution

class Pointer:
    __slots__ = ("cell", "path")

    def __init__(self, cell, path=()):
        self.cell, self.path = cell, path

    def load(self):
        v = self.cell[0]
        for k in self.path:
            v = v[k]
        return v

    def store(self, val):
        if not self.path:
            self.cell[0] = val
            return
        v = self.cell[0]
        for k in self.path[:-1]:
            v = v[k]
        v[self.path[-1]] = val   # in place: vectors are numpy arrays, aggregates are lists


def _f(x):
    return np.asarray(x, dtype=np.float32) if isinstance(x, np.ndarray) else F(x)


class Shader:
    """One fragment shader.  run(inputs, resources) executes main() and returns the outputs.

    inputs:    {location: value}, plus "frag_coord" for gl_FragCoord
    resources: {(set, binding): Tex2D | TexCube}, plus "push" = list of push-constant member values
    """

    def __init__(self, path, nan_max="propagate"):
        """nan_max: what max / min / clamp return when an operand is NaN, which GLSL leaves undefined: "propagate" (NaN,
        numpy's maximum) or "maxnum" (the other operand, IEEE 754-2008 maxNum, what v_max_f32 and most GPUs do)."""
        self.m = Module(path)
        self.ops = 0
        self._max, self._min = (np.fmax, np.fmin) if nan_max == "maxnum" else (np.maximum, np.minimum)

    # ---- helpers ----
    def _default(self, ty):
        t = self.m.types[ty]
        if t[0] == "float":
            return F(0)
        if t[0] == "int":
            return np.int32(0) if t[2] else np.uint32(0)
        if t[0] == "bool":
            return False
        if t[0] == "vec":
            return np.zeros(t[2], dtype=np.float32)
        if t[0] == "mat":
            return [np.zeros(self.m.types[t[1]][2], dtype=np.float32) for _ in range(t[2])]
        if t[0] == "array":
            return [self._default(t[1]) for _ in range(int(self.m.const_value(t[2])))]
        if t[0] == "struct":
            return [self._default(x) for x in t[1]]
        return None

    def run(self, inputs, resources):
        m = self.m
        self.globals = {}
        outputs = {}
        for gid, g in m.globals.items():
            pt = m.types[g["type"]]
            dec = m.decor.get(gid, {})
            sc = g["storage"]
            if sc == 1:      # Input
                if 11 in dec and dec[11][0] == 15:   # BuiltIn FragCoord
                    val = np.asarray(inputs["frag_coord"], dtype=np.float32)
                elif 30 not in dec:                  # gl_in[] of a geometry shader: list of gl_PerVertex members
                    self.globals[gid] = Pointer([inputs["gl_in"]])
                    continue
                else:
                    val = inputs[dec[30][0]]
                    t = m.types[pt[2]]
                    if t[0] == "mat":
                        val = [np.asarray(c, dtype=np.float32) for c in val]
                    elif t[0] == "vec":
                        val = np.asarray(val, dtype=np.float32)
                    else:
                        val = F(val)
                self.globals[gid] = Pointer([val])
            elif sc == 3:    # Output
                cell = [self._default(pt[2])]
                self.globals[gid] = Pointer(cell)
                # no Location: the gl_PerVertex block, or a single built-in such as gl_Layer (BuiltIn 9)
                outputs[dec[30][0] if 30 in dec else (("builtin", dec[11][0]) if 11 in dec else "builtin")] = cell
            elif sc == 2:    # Uniform (a uniform block: list of members, matrices as lists of columns)
                self.globals[gid] = Pointer([list(resources[(dec[34][0], dec[33][0])])])
            elif sc == 0:    # UniformConstant (samplers)
                self.globals[gid] = Pointer([resources[(dec[34][0], dec[33][0])]])
            elif sc == 9:    # PushConstant
                self.globals[gid] = Pointer([list(resources["push"])])
            else:            # Private
                init = m.const_value(g["init"]) if g["init"] is not None else self._default(pt[2])
                self.globals[gid] = Pointer([init])
        killed = False
        self.emitted = []          # geometry shaders: a copy of every output at each EmitVertex
        self._outputs = outputs
        try:
            self._call(m.entry, [])
        except Kill:
            killed = True
        return {"killed": killed, "out": {loc: cell[0] for loc, cell in outputs.items()}}

    def _call(self, fid, args):
        m = self.m
        fn = m.functions[fid]
        env = dict(zip(fn["params"], args))

        def val(i):
            if i in env:
                return env[i]
            if i in self.globals:
                return self.globals[i]
            return m.const_value(i)

        label = fn["order"][0]
        while True:
            for op, a in fn["blocks"][label]:
                self.ops += 1
                if op == 59:     # OpVariable (Function storage)
                    pt = m.types[a[0]]
                    init = m.const_value(a[3]) if len(a) > 3 else self._default(pt[2])
                    env[a[1]] = Pointer([init])
                elif op == 61:   # Load
                    v = val(a[2]).load()
                    env[a[1]] = v.copy() if isinstance(v, np.ndarray) else v
                elif op == 62:   # Store
                    v = val(a[1])
                    val(a[0]).store(v.copy() if isinstance(v, np.ndarray) else v)
                elif op == 65:   # AccessChain
                    base = val(a[2])
                    idx = tuple(int(val(k)) for k in a[3:])
                    env[a[1]] = Pointer(base.cell, base.path + idx)
                elif op == 57:   # FunctionCall
                    env[a[1]] = self._call(a[2], [val(k) for k in a[3:]])
                elif op == 12:   # ExtInst
                    env[a[1]] = self._ext(a[3], [val(k) for k in a[4:]])
                elif op == 79:   # VectorShuffle
                    both = np.concatenate([np.atleast_1d(val(a[2])), np.atleast_1d(val(a[3]))])
                    env[a[1]] = both[list(a[4:])].astype(np.float32)
                elif op == 80:   # CompositeConstruct
                    t = m.types[a[0]]
                    parts = [val(k) for k in a[2:]]
                    if t[0] == "vec":
                        env[a[1]] = np.concatenate([np.atleast_1d(p) for p in parts]).astype(np.float32)
                    else:
                        env[a[1]] = list(parts)
                elif op == 81:   # CompositeExtract
                    v = val(a[2])
                    for k in a[3:]:
                        v = v[k]
                    env[a[1]] = F(v) if isinstance(v, (np.floating, float)) else v
                elif op in (87, 88):   # ImageSample{Implicit,Explicit}Lod (every lod resolves to level 0)
                    env[a[1]] = np.asarray(val(a[2]).sample(val(a[3])), dtype=np.float32)
                elif op == 103:  # ImageQuerySizeLod
                    env[a[1]] = np.array(val(a[2]).size(), dtype=np.int64)
                elif op == 100:  # OpImage
                    env[a[1]] = val(a[2])
                elif op in (111, 112):
                    v = val(a[2])
                    env[a[1]] = v.astype(np.float32) if isinstance(v, np.ndarray) else F(v)
                elif op == 127:
                    env[a[1]] = -val(a[2])
                elif op == 128:
                    x, y = val(a[2]), val(a[3])
                    if isinstance(x, np.uint32) or isinstance(y, np.uint32):
                        env[a[1]] = np.uint32((int(x) + int(y)) & 0xFFFFFFFF)
                    else:
                        env[a[1]] = np.int32(int(x) + int(y))
                elif op == 129:
                    env[a[1]] = _f(val(a[2]) + val(a[3]))
                elif op == 131:
                    env[a[1]] = _f(val(a[2]) - val(a[3]))
                elif op == 133:
                    env[a[1]] = _f(val(a[2]) * val(a[3]))
                elif op == 136:
                    with np.errstate(divide="ignore", invalid="ignore"):
                        env[a[1]] = _f(val(a[2]) / val(a[3]))
                elif op == 142:  # VectorTimesScalar
                    env[a[1]] = _f(val(a[2]) * val(a[3]))
                elif op == 145:  # MatrixTimesVector: sum of columns scaled by the vector's components
                    cols, v = val(a[2]), val(a[3])
                    acc = cols[0] * v[0]
                    for c in range(1, len(cols)):
                        acc = _f(acc + _f(cols[c] * v[c]))
                    env[a[1]] = _f(acc)
                elif op == 146:  # MatrixTimesMatrix: column j of the result = left * (column j of right)
                    left, right = val(a[2]), val(a[3])
                    res = []
                    for rc in right:
                        acc = left[0] * rc[0]
                        for c in range(1, len(left)):
                            acc = _f(acc + _f(left[c] * rc[c]))
                        res.append(_f(acc))
                    env[a[1]] = res
                elif op == 148:  # Dot
                    x, y = val(a[2]), val(a[3])
                    acc = F(x[0] * y[0])
                    for c in range(1, len(x)):
                        acc = F(acc + F(x[c] * y[c]))
                    env[a[1]] = acc
                elif op == 169:  # Select
                    c, x, y = val(a[2]), val(a[3]), val(a[4])
                    env[a[1]] = np.where(c, x, y) if isinstance(c, np.ndarray) else (x if c else y)
                elif op == 171:
                    env[a[1]] = bool(int(val(a[2])) != int(val(a[3])))
                elif op in (176, 177):
                    env[a[1]] = bool(int(val(a[2])) < int(val(a[3])))
                elif op == 179:
                    env[a[1]] = bool(int(val(a[2])) <= int(val(a[3])))
                elif op == 180:
                    env[a[1]] = bool(val(a[2]) == val(a[3]))
                elif op == 184:
                    env[a[1]] = bool(val(a[2]) < val(a[3]))
                elif op == 186:
                    env[a[1]] = bool(val(a[2]) > val(a[3]))
                elif op == 194:
                    env[a[1]] = np.uint32(int(val(a[2])) >> int(val(a[3])))
                elif op == 196:
                    env[a[1]] = np.uint32((int(val(a[2])) << int(val(a[3]))) & 0xFFFFFFFF)
                elif op == 197:
                    env[a[1]] = np.uint32(int(val(a[2])) | int(val(a[3])))
                elif op == 199:
                    env[a[1]] = np.uint32(int(val(a[2])) & int(val(a[3])))
                elif op in (246, 247):
                    pass  # structured-control-flow annotations
                elif op == 249:
                    label = a[0]
                    break
                elif op == 250:
                    label = a[1] if val(a[0]) else a[2]
                    break
                elif op == 252:
                    raise Kill()
                elif op == 218:    # OpEmitVertex
                    import copy
                    self.emitted.append({loc: copy.deepcopy(cell[0]) for loc, cell in self._outputs.items()})
                elif op == 219:    # OpEndPrimitive
                    self.emitted.append(None)
                elif op == 253:
                    return None
                elif op == 254:
                    return val(a[0])
                else:
                    raise NotImplementedError(f"SPIR-V opcode {op}")
            else:
                raise RuntimeError("block fell through without a terminator")

    def _ext(self, inst, x):
        with np.errstate(divide="ignore", invalid="ignore"):
            if inst == 4:
                return _f(np.abs(x[0]))
            if inst == 13:
                return _f(np.sin(x[0]))
            if inst == 14:
                return _f(np.cos(x[0]))
            if inst == 16:
                return _f(np.arcsin(x[0]))
            if inst == 25:
                return _f(np.arctan2(x[0], x[1]))
            if inst == 26:
                return _f(np.power(x[0], x[1]))
            if inst == 30:
                return _f(np.log2(x[0]))
            if inst == 31:
                return _f(np.sqrt(x[0]))
            if inst == 40:
                return _f(self._max(x[0], x[1]))
            if inst == 43:
                return _f(self._min(self._max(x[0], x[1]), x[2]))
            if inst == 46:   # mix(x, y, a) = x*(1-a) + y*a
                return _f(_f(x[0] * _f(F(1) - x[2])) + _f(x[1] * x[2]))
            if inst == 66:
                return F(np.sqrt(self._dot(x[0], x[0])))
            if inst == 68:
                a, b = x[0], x[1]
                return np.array([F(a[1] * b[2]) - F(a[2] * b[1]), F(a[2] * b[0]) - F(a[0] * b[2]), F(a[0] * b[1]) - F(a[1] * b[0])], dtype=np.float32)
            if inst == 69:
                return _f(x[0] / F(np.sqrt(self._dot(x[0], x[0]))))
            if inst == 71:   # reflect(I, N) = I - 2 dot(N, I) N
                return _f(x[0] - _f(_f(F(2) * self._dot(x[1], x[0])) * x[1]))
        raise NotImplementedError(f"GLSL.std.450 instruction {inst}")

    @staticmethod
    def _dot(x, y):
        acc = F(x[0] * y[0])
        for c in range(1, len(x)):
            acc = F(acc + F(x[c] * y[c]))
        return acc
