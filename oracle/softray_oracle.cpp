/*
 * softray_oracle.cpp -- CPU ORACLE: statement-level restatement of the raytrace hot path of
 * voidstar69/softray.  TEST INFRASTRUCTURE ONLY (see softray_oracle.h).
 *
 * Build: g++ -std=c++17 -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (oracle/Makefile).
 * All arithmetic is IEEE double in the operation order of the C# expressions; no FMA.
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 *
 * Third-party arithmetic that is NOT in the reference tree (un-vendored .NET Framework 4.0 BCL):
 * System.Random (Knuth subtractive generator) and System.Math.{Sqrt,Pow,Sin,Cos,Tan}.  Random is
 * restated from its published algorithm and pinned by the reference's own
 * seeded tests; Math.* maps to glibc libm.
 */
#include "softray_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cfloat>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <thread>
#include <atomic>
#include <algorithm>

namespace {

/* =========================================================================================
 * System.Random  (.NET Framework 4.0 BCL; not under /root/reference)
 * call sites: Renderer.cs:1624,1693  ShadowMethod.cs:63-73  SpatialSubdivisionTests.cs:64,397-411
 * ========================================================================================= */
struct DotNetRandom {
    static constexpr int32_t MBIG = 2147483647;
    static constexpr int32_t MSEED = 161803398;
    int32_t seedArray[56];
    int inext, inextp;

    explicit DotNetRandom(int32_t seed) {
        int32_t subtraction = (seed == INT32_MIN) ? INT32_MAX : std::abs(seed);
        int32_t mj = MSEED - subtraction;
        seedArray[55] = mj;
        int32_t mk = 1;
        seedArray[0] = 0;
        for (int i = 1; i < 55; i++) {
            int ii = (21 * i) % 55;
            seedArray[ii] = mk;
            mk = mj - mk;
            if (mk < 0) mk += MBIG;
            mj = seedArray[ii];
        }
        for (int k = 1; k < 5; k++) {
            for (int i = 1; i < 56; i++) {
                /* C# int arithmetic wraps; do it in uint32 to stay defined in C++ */
                seedArray[i] = (int32_t)((uint32_t)seedArray[i] - (uint32_t)seedArray[1 + (i + 30) % 55]);
                if (seedArray[i] < 0) seedArray[i] += MBIG;
            }
        }
        inext = 0;
        inextp = 21;
    }
    int32_t InternalSample() {
        int locINext = inext, locINextp = inextp;
        if (++locINext >= 56) locINext = 1;
        if (++locINextp >= 56) locINextp = 1;
        int32_t retVal = seedArray[locINext] - seedArray[locINextp];
        if (retVal == MBIG) retVal--;
        if (retVal < 0) retVal += MBIG;
        seedArray[locINext] = retVal;
        inext = locINext;
        inextp = locINextp;
        return retVal;
    }
    double Sample() { return InternalSample() * (1.0 / MBIG); }
    int32_t Next() { return InternalSample(); }
    int32_t Next(int32_t maxValue) { return (int32_t)(Sample() * maxValue); }
    double NextDouble() { return Sample(); }
};

/* =========================================================================================
 * Vector  (Engine3D/Vector.cs:9-197)
 * ========================================================================================= */
struct Vec {
    double x, y, z;
};
inline Vec V(double x, double y, double z) { return Vec{x, y, z}; }
inline Vec operator+(Vec a, Vec b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }  /* :55 */
inline Vec operator-(Vec a, Vec b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }  /* :60 */
inline Vec operator*(Vec a, double b) { return V(a.x * b, a.y * b, a.z * b); }     /* :69 */
inline Vec operator*(double a, Vec b) { return V(a * b.x, a * b.y, a * b.z); }     /* :74 */
inline Vec operator-(Vec v) { return V(-v.x, -v.y, -v.z); }                        /* :92 */
inline double Dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      /* :99 */
inline Vec Cross(Vec a, Vec b) {                                                   /* :104 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline double Length(Vec v) { return std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); } /* :121 */
inline double Distance(Vec a, Vec b) { return Length(a - b); }                      /* :112 */
inline bool IsZeroVector(Vec v) {                                                  /* :140 */
    const double epsilon = 1e-10;
    return -epsilon < v.x && v.x < epsilon && -epsilon < v.y && v.y < epsilon && -epsilon < v.z && v.z < epsilon;
}
inline void Normalise(Vec& v) {                                                    /* :177-185 */
    double len = Length(v);
    double inverseLen = 1.0 / len;
    v.x *= inverseLen;
    v.y *= inverseLen;
    v.z *= inverseLen;
}

/* C# unchecked (byte)(double): truncate toward zero, keep low 8 bits. NaN / out-of-int-range are
 * undefined in C#; x64 cvttsd2si yields 0x80000000 -> low byte 0. */
inline uint8_t ToByte(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return 0;
    return (uint8_t)(int32_t)d;
}

/* Color.ModulatePackedColor (Engine3D/Color.cs:124-133) */
inline uint32_t ModulatePackedColor(uint32_t color, uint8_t amount) {
    uint8_t r = (uint8_t)(color >> 16);
    uint8_t g = (uint8_t)(color >> 8);
    uint8_t b = (uint8_t)color;
    r = (uint8_t)((r * amount) >> 8);
    g = (uint8_t)((g * amount) >> 8);
    b = (uint8_t)((b * amount) >> 8);
    return (255u << 24) + ((uint32_t)r << 16) + ((uint32_t)g << 8) + b;
}
/* Color.ToARGB (Color.cs:105-111) */
inline uint32_t ColorToARGB(double r, double g, double b) {
    return (255u << 24) + ((uint32_t)ToByte(r * 255.0) << 16) + ((uint32_t)ToByte(g * 255.0) << 8) + ToByte(b * 255.0);
}
/* Surface.PackColorAndAlpha (Engine3D/Surface.cs:131-138) */
inline uint32_t PackColorAndAlpha(double r, double g, double b, double alpha) {
    uint8_t rb = ToByte(r * 255.0), gb = ToByte(g * 255.0), bb = ToByte(b * 255.0), ab = ToByte(alpha * 255.0);
    return ((uint32_t)ab << 24) + ((uint32_t)rb << 16) + ((uint32_t)gb << 8) + bb;
}

/* =========================================================================================
 * IntersectionInfo (Engine3D/Raytrace/IRayIntersectable.cs:5-18)
 * ========================================================================================= */
struct Hit {
    bool valid = false;
    double rayFrac = 0;
    Vec pos{0, 0, 0};
    Vec normal{0, 0, 0};
    uint32_t color = 0;
    int triIndex = -1;
};

/* per-ray counters: the reference's NumRayTests / NumNodesVisited / NumLeafNodesVisited */
struct Counters {
    int64_t geomTests = 0, nodeVisits = 0, leafVisits = 0;
};

/* =========================================================================================
 * Plane (Engine3D/Raytrace/Plane.cs)
 * ========================================================================================= */
struct Plane {
    Vec normal;
    double originDist;
    uint32_t color;
    Plane() : normal{1, 0, 0}, originDist(0), color(0xffffffffu) {}
    Plane(Vec point, Vec n) {                                /* Plane.cs:22-29 */
        normal = n;
        Normalise(normal);
        originDist = Dot(point, normal);
        color = ColorToARGB(1.0, 1.0, 1.0);                  /* Color = Color.White */
    }
    /* Plane.cs:67-103 */
    bool IntersectRay(Vec start, Vec dir, Hit& info) const {
        double startDist = Dot(start, normal);
        double dirDist = Dot(dir, normal);
        if (dirDist >= 0.0) return false;                    /* one-sided */
        double rayFrac = originDist - startDist;
        if (rayFrac <= 0.0) {
            rayFrac /= dirDist;
            info.valid = true;
            info.pos = start + dir * rayFrac;
            info.normal = normal;
            info.rayFrac = rayFrac;
            info.color = color;
            info.triIndex = -1;
            return true;
        }
        return false;
    }
    /* Plane.cs:111-138 */
    bool IntersectLineSegment(Vec start, Vec end, double& lineFracOut, Vec& posOut) const {
        double startDist = Dot(start, normal);
        double endDist = Dot(end, normal);
        double lineFrac = (originDist - startDist) / (endDist - startDist);
        if (0.0 <= lineFrac && lineFrac <= 1.0) {
            lineFracOut = lineFrac;
            posOut = start + (end - start) * lineFrac;
            return true;
        }
        return false;
    }
};

/* Point.IntersectPlane (Engine3D/Raytrace/Point.cs:35-50): true = NormalSide */
inline bool PointOnNormalSide(Vec pos, const Plane& plane) {
    double distToOriginAlongNormal = Dot(pos, plane.normal);
    return distToOriginAlongNormal >= plane.originDist;
}

/* =========================================================================================
 * Triangle (Engine3D/Raytrace/Triangle.cs)
 * ========================================================================================= */
struct Triangle {
    Plane plane;
    Vec v1, v2, v3;
    Vec edge1, edge2;
    Vec edge1Perp, edge2Perp;
    uint32_t color;
    int triangleIndex;

    Triangle(Vec a, Vec b, Vec c, uint32_t col) {            /* Triangle.cs:29-57 */
        v1 = a; v2 = b; v3 = c;
        color = col;
        edge1 = v2 - v1;
        edge2 = v3 - v1;
        Vec normal = Cross(edge1, edge2);
        if (IsZeroVector(normal)) normal = V(1, 0, 0);
        plane = Plane(v1, normal);
        edge1Perp = Cross(edge1, normal);
        edge2Perp = Cross(edge2, normal);
        triangleIndex = -1;
    }
    /* Triangle.cs:83-104 */
    bool IntersectRay(Vec start, Vec dir, Hit& info) const {
        if (!plane.IntersectRay(start, dir, info)) return false;
        Vec v1ToIntersection = info.pos - v1;
        double s = Dot(v1ToIntersection, edge2Perp) / Dot(edge1, edge2Perp);
        if (s < 0.0 || s > 1.0) { info.valid = false; return false; }
        double t = Dot(v1ToIntersection, edge1Perp) / Dot(edge2, edge1Perp);
        if (s >= 0.0 && t >= 0.0 && s + t <= 1.0) {
            info.color = color;
            info.triIndex = triangleIndex;
            return true;
        }
        info.valid = false;
        return false;
    }
    /* Triangle.cs:125-131 : bit0 = NormalSide, bit1 = BackSide */
    int IntersectPlane(const Plane& p) const {
        int h = 0;
        h |= PointOnNormalSide(v1, p) ? 1 : 2;
        h |= PointOnNormalSide(v2, p) ? 1 : 2;
        h |= PointOnNormalSide(v3, p) ? 1 : 2;
        return h;
    }
};

/* =========================================================================================
 * Sphere.IntersectRay (Engine3D/Raytrace/Sphere.cs:152-219)
 * ========================================================================================= */
struct Sphere {
    Vec center;
    double radius, radiusSqr;
    uint32_t color;
    Sphere(Vec c, double r, uint32_t col) : center(c), radius(r), radiusSqr(r * r), color(col) {}
    bool IntersectRay(Vec start, Vec dir, Hit& info) const {
        const double epsilon = 1e-10;
        Normalise(dir);
        Vec sphereToStart = start - center;
        double sphereToStartProjDir = Dot(sphereToStart, dir);
        if (sphereToStartProjDir > radius) return false;
        double sphereToStartDistSqr = sphereToStart.x * sphereToStart.x + sphereToStart.y * sphereToStart.y + sphereToStart.z * sphereToStart.z;
        double termUnderSqrRoot = sphereToStartProjDir * sphereToStartProjDir - sphereToStartDistSqr + radiusSqr;
        if (termUnderSqrRoot < epsilon) return false;
        double positiveSqrRoot = std::sqrt(termUnderSqrRoot);
        double intersectRayFrac1 = -sphereToStartProjDir - positiveSqrRoot;
        double intersectRayFrac2 = -sphereToStartProjDir + positiveSqrRoot;
        double rayFrac = (intersectRayFrac1 >= 0 ? intersectRayFrac1 : intersectRayFrac2);
        if (rayFrac < 0) return false;
        info.valid = true;
        info.rayFrac = rayFrac;
        info.pos = start + dir * rayFrac;
        info.normal = info.pos - center;
        Normalise(info.normal);
        info.color = color;
        info.triIndex = -1;
        return true;
    }
};

/* =========================================================================================
 * AxisAlignedBox (Engine3D/Raytrace/AxisAlignedBox.cs)
 * ========================================================================================= */
struct AxisAlignedBox {
    Vec min, max;
    Plane planes[6];
    AxisAlignedBox() : min{0, 0, 0}, max{0, 0, 0} {}
    AxisAlignedBox(Vec mn, Vec mx) : min(mn), max(mx) {      /* :16-28 */
        planes[0] = Plane(min, V(-1, 0, 0));
        planes[1] = Plane(min, V(0, -1, 0));
        planes[2] = Plane(min, V(0, 0, -1));
        planes[3] = Plane(max, V(+1, 0, 0));
        planes[4] = Plane(max, V(0, +1, 0));
        planes[5] = Plane(max, V(0, 0, +1));
    }
    Vec Centre() const { return (min + max) * 0.5; }         /* :46-52 */
    bool ContainsPoint(Vec pos) const {                      /* :143-149 */
        const double epsilon = 1e-10;
        return min.x - epsilon < pos.x && pos.x < max.x + epsilon &&
               min.y - epsilon < pos.y && pos.y < max.y + epsilon &&
               min.z - epsilon < pos.z && pos.z < max.z + epsilon;
    }
    /* AxisAlignedBox.IntersectRay, :60-95: the nearest of the six one-sided planes' hits that lies on the box (ContainsPoint) */
    bool IntersectRay(Vec start, Vec dir, Hit& info) const {
        Hit closest;
        closest.rayFrac = DBL_MAX;
        for (int i = 0; i < 6; i++) {
            Hit curr;
            if (planes[i].IntersectRay(start, dir, curr) && curr.rayFrac < closest.rayFrac) {
                if (ContainsPoint(curr.pos)) closest = curr;
            }
        }
        if (closest.rayFrac == DBL_MAX) return false;
        info = closest;
        return true;
    }
    /* :111-141 */
    bool IntersectLineSegment(Vec start, Vec end, Vec& posOut) const {
        double closest = DBL_MAX;
        Vec closestPos{0, 0, 0};
        for (int i = 0; i < 6; i++) {
            double f; Vec p;
            if (planes[i].IntersectLineSegment(start, end, f, p) && f < closest) {
                if (ContainsPoint(p)) { closest = f; closestPos = p; }
            }
        }
        if (closest == DBL_MAX) return false;
        posOut = closestPos;
        return true;
    }
    /* :175-216 */
    bool ClipLineSegment(Vec& start, Vec& end) const {
        bool startInside = ContainsPoint(start);
        bool endInside = ContainsPoint(end);
        if (startInside && endInside) return true;
        Vec ipos;
        if (!IntersectLineSegment(start, end, ipos)) return false;
        if (startInside) { end = ipos; return true; }
        Vec originalStart = start;
        start = ipos;
        if (!endInside) {
            /* Contract.Assume(intersection != null): if it were null the C# would throw a
             * NullReferenceException; keep the segment end unchanged in that (unobserved) case. */
            if (IntersectLineSegment(end, originalStart, ipos)) end = ipos;
        }
        return true;
    }
};

/* =========================================================================================
 * SpatialSubdivision (Engine3D/Raytrace/SpatialSubdivision.cs)
 * ========================================================================================= */
struct Node {
    std::vector<int> geometry;   /* indices into the triangle list, list order preserved */
    AxisAlignedBox boundingBox;
    Plane splittingPlane;
    bool hasPlane = false;
    std::unique_ptr<Node> normalSide, backSide;
};

struct TreeStats { int totalTreeDepth = 0, totalNodes = 0, leafNodes = 0; };

struct SpatialSubdivision {
    std::unique_ptr<Node> root;
    const std::vector<Triangle>* tris = nullptr;
    int TreeDepth = 0, NumNodes = 0, NumLeafNodes = 0, NumInternalNodes = 0;

    /* Node.RecursivePlaneSplit (:49-230) */
    static void RecursivePlaneSplit(Node* node, const std::vector<Triangle>& T, TreeStats& st,
                                    int treeDepth, int maxTreeDepth, int maxGeometryPerNode) {
        st.totalTreeDepth = std::max(st.totalTreeDepth, treeDepth);
        if (treeDepth >= maxTreeDepth || (int)node->geometry.size() <= maxGeometryPerNode) {
            st.leafNodes++;
            return;
        }
        int axis;
        Vec boxExtent = node->boundingBox.max - node->boundingBox.min;
        boxExtent = V(std::fabs(boxExtent.x), std::fabs(boxExtent.y), std::fabs(boxExtent.z));
        if (boxExtent.x > boxExtent.y) {
            if (boxExtent.x > boxExtent.z) axis = 0; else axis = 2;
        } else {
            if (boxExtent.y > boxExtent.z) axis = 1; else axis = 2;
        }
        Vec splitPt = node->boundingBox.Centre();
        switch (axis) {
            case 0: node->splittingPlane = Plane(splitPt, V(1, 0, 0)); break;
            case 1: node->splittingPlane = Plane(splitPt, V(0, 1, 0)); break;
            default: node->splittingPlane = Plane(splitPt, V(0, 0, 1)); break;
        }
        node->hasPlane = true;
        std::vector<int> normalSideGeom, backSideGeom;
        for (int gi : node->geometry) {
            int h = T[gi].IntersectPlane(node->splittingPlane);
            if (h & 1) normalSideGeom.push_back(gi);
            if (h & 2) backSideGeom.push_back(gi);
        }
        bool rejectSplit = (normalSideGeom.size() == node->geometry.size() || backSideGeom.size() == node->geometry.size());
        if (rejectSplit) {
            node->hasPlane = false;
            st.leafNodes++;
            return;
        }
        node->geometry.clear();
        node->geometry.shrink_to_fit();
        Vec backSideBoxMax = node->boundingBox.max;
        Vec normalSideBoxMin = node->boundingBox.min;
        switch (axis) {
            case 0: backSideBoxMax.x = normalSideBoxMin.x = splitPt.x; break;
            case 1: backSideBoxMax.y = normalSideBoxMin.y = splitPt.y; break;
            default: backSideBoxMax.z = normalSideBoxMin.z = splitPt.z; break;
        }
        AxisAlignedBox backSideBox(node->boundingBox.min, backSideBoxMax);
        AxisAlignedBox normalSideBox(normalSideBoxMin, node->boundingBox.max);
        treeDepth++;
        if (!normalSideGeom.empty()) {
            node->normalSide.reset(new Node());
            node->normalSide->geometry = std::move(normalSideGeom);
            node->normalSide->boundingBox = normalSideBox;
            st.totalNodes++;
            RecursivePlaneSplit(node->normalSide.get(), T, st, treeDepth, maxTreeDepth, maxGeometryPerNode);
        }
        if (!backSideGeom.empty()) {
            node->backSide.reset(new Node());
            node->backSide->geometry = std::move(backSideGeom);
            node->backSide->boundingBox = backSideBox;
            st.totalNodes++;
            RecursivePlaneSplit(node->backSide.get(), T, st, treeDepth, maxTreeDepth, maxGeometryPerNode);
        }
        if (!node->normalSide && !node->backSide) {          /* ":TODO we never seem to reach this" */
            st.leafNodes++;
            node->hasPlane = false;
        }
    }

    /* ctor (:267-315). returns false if a vertex is outside the box (ArgumentOutOfRangeException :293) */
    bool Build(const std::vector<Triangle>& T, const AxisAlignedBox& box, int maxTreeDepth, int maxGeometryPerNode) {
        tris = &T;
        for (const Triangle& tri : T) {
            if (!box.ContainsPoint(tri.v1) || !box.ContainsPoint(tri.v2) || !box.ContainsPoint(tri.v3)) return false;
        }
        TreeStats st;
        root.reset(new Node());
        root->geometry.resize(T.size());
        for (size_t i = 0; i < T.size(); i++) root->geometry[i] = (int)i;
        root->boundingBox = box;
        st.totalNodes++;
        RecursivePlaneSplit(root.get(), T, st, 1, maxTreeDepth, maxGeometryPerNode);
        TreeDepth = st.totalTreeDepth;
        NumNodes = st.totalNodes;
        NumLeafNodes = st.leafNodes;
        NumInternalNodes = NumNodes - NumLeafNodes;
        return true;
    }

    /* GetClosestIntersection (:629-676).  The testedTriangles HashSet never changes a result of
     * IntersectRay (a leaf that accepts a hit ends the traversal) and is dropped; SURVEY.md 8a row L. */
    Hit GetClosestIntersection(const Node* node, Vec start, Vec dir, Counters& c) const {
        Hit closest;
        closest.rayFrac = DBL_MAX;
        for (int gi : node->geometry) {
            Hit h;
            if ((*tris)[gi].IntersectRay(start, dir, h) && h.rayFrac < closest.rayFrac) {
                if (node->boundingBox.ContainsPoint(h.pos)) closest = h;
            }
            c.geomTests++;
        }
        return closest;
    }

    /* RecursiveRayTrace (:458-627) */
    bool RecursiveRayTrace(const Node* node, Vec start, Vec end, Vec dir, Hit& out, Counters& c) const {
        if (node == nullptr) return false;
        c.nodeVisits++;
        if (!node->normalSide && !node->backSide) {
            c.leafVisits++;
            Hit closest = GetClosestIntersection(node, start, dir, c);
            if (closest.rayFrac < DBL_MAX) { out = closest; return true; }
            return false;
        }
        bool startNormal = PointOnNormalSide(start, node->splittingPlane);
        bool endNormal = PointOnNormalSide(end, node->splittingPlane);
        if (startNormal) {
            if (RecursiveRayTrace(node->normalSide.get(), start, end, dir, out, c)) return true;
            if (!endNormal) return RecursiveRayTrace(node->backSide.get(), start, end, dir, out, c);
        } else {
            if (RecursiveRayTrace(node->backSide.get(), start, end, dir, out, c)) return true;
            if (endNormal) return RecursiveRayTrace(node->normalSide.get(), start, end, dir, out, c);
        }
        return false;
    }

    /* IntersectRay (:381-419) */
    bool IntersectRay(Vec start, Vec dir, Hit& out, Counters& c) const {
        Vec end = start + dir * 10000;
        Vec originalStart = start;
        if (!root->boundingBox.ClipLineSegment(start, end)) return false;
        double rayFracOffset = Distance(originalStart, start) / Length(dir);
        if (RecursiveRayTrace(root.get(), start, end, dir, out, c)) {
            out.rayFrac += rayFracOffset;
            return true;
        }
        return false;
    }
};

/* =========================================================================================
 * Matrix / Instance (Engine3D/Matrix.cs, Engine3D/Instance.cs)
 * ========================================================================================= */
struct Matrix {
    double m[4][4];
    Matrix() { std::memset(m, 0, sizeof(m)); }
};
inline Matrix operator*(const Matrix& m1, const Matrix& m2) {   /* Matrix.cs:74-91 */
    Matrix r;
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++) {
            double sum = 0.0;
            for (int i = 0; i < 4; i++) sum += m1.m[row][i] * m2.m[i][col];
            r.m[row][col] = sum;
        }
    return r;
}
Matrix MakeTranslationMatrix(Vec p) {                            /* Matrix.cs:94-113 */
    Matrix t;
    t.m[0][0] = 1.0; t.m[1][1] = 1.0; t.m[2][2] = 1.0;
    t.m[0][3] = p.x; t.m[1][3] = p.y; t.m[2][3] = p.z; t.m[3][3] = 1.0;
    return t;
}
Matrix MakeYawMatrix(double yaw) {                               /* Matrix.cs:116-131 */
    Matrix y;
    y.m[0][0] = std::cos(yaw); y.m[1][0] = 0.0; y.m[2][0] = std::sin(yaw);
    y.m[0][1] = 0.0; y.m[1][1] = 1.0; y.m[2][1] = 0.0;
    y.m[0][2] = -std::sin(yaw); y.m[1][2] = 0.0; y.m[2][2] = std::cos(yaw);
    y.m[3][3] = 1.0;
    return y;
}
Matrix MakePitchMatrix(double pitch) {                           /* Matrix.cs:134-149 */
    Matrix p;
    p.m[0][0] = 1.0; p.m[1][0] = 0.0; p.m[2][0] = 0.0;
    p.m[0][1] = 0.0; p.m[1][1] = std::cos(pitch); p.m[2][1] = std::sin(pitch);
    p.m[0][2] = 0.0; p.m[1][2] = -std::sin(pitch); p.m[2][2] = std::cos(pitch);
    p.m[3][3] = 1.0;
    return p;
}
Matrix MakeRollMatrix(double roll) {                             /* Matrix.cs:152-167 */
    Matrix r;
    r.m[0][0] = std::cos(roll); r.m[1][0] = std::sin(roll); r.m[2][0] = 0.0;
    r.m[0][1] = -std::sin(roll); r.m[1][1] = std::cos(roll); r.m[2][1] = 0.0;
    r.m[0][2] = 0.0; r.m[1][2] = 0.0; r.m[2][2] = 1.0;
    r.m[3][3] = 1.0;
    return r;
}

/* the per-frame view of an Instance: rows 0..2 of _transform / _inverseTransform */
struct InstanceXf {
    double t[3][4], it[3][4];
    double positionZ, fovDepth;
    /* Matrix.Multiply3X4 (Matrix.cs:49-56) + projection, Instance.TransformPosToView (Instance.cs:168-184) */
    Vec TransformPosToView(Vec pos) const {
        Vec v = V(pos.x * t[0][0] + pos.y * t[0][1] + pos.z * t[0][2] + t[0][3],
                  pos.x * t[1][0] + pos.y * t[1][1] + pos.z * t[1][2] + t[1][3],
                  pos.x * t[2][0] + pos.y * t[2][1] + pos.z * t[2][2] + t[2][3]);
        v.x = v.x / v.z * fovDepth;
        v.y = v.y / v.z * fovDepth;
        v.z = (v.z - positionZ + 1.0) * 0.5;
        return v;
    }
    /* Instance.TransformPosFromView (Instance.cs:192-209): the un-projection is computed and then
     * IGNORED -- the method returns _inverseTransform(3x4) * pos.  Bug kept. */
    Vec TransformPosFromView(Vec pos) const {
        return V(pos.x * it[0][0] + pos.y * it[0][1] + pos.z * it[0][2] + it[0][3],
                 pos.x * it[1][0] + pos.y * it[1][1] + pos.z * it[1][2] + it[1][3],
                 pos.x * it[2][0] + pos.y * it[2][1] + pos.z * it[2][2] + it[2][3]);
    }
    Vec TransformDirection(Vec d) const {                        /* Instance.cs:216-222 */
        return V(d.x * t[0][0] + d.y * t[0][1] + d.z * t[0][2],
                 d.x * t[1][0] + d.y * t[1][1] + d.z * t[1][2],
                 d.x * t[2][0] + d.y * t[2][1] + d.z * t[2][2]);
    }
    Vec TransformDirectionReverse(Vec d) const {                 /* Instance.cs:229-235 */
        return V(d.x * it[0][0] + d.y * it[0][1] + d.z * it[0][2],
                 d.x * it[1][0] + d.y * it[1][1] + d.z * it[1][2],
                 d.x * it[2][0] + d.y * it[2][1] + d.z * it[2][2]);
    }
};

/* =========================================================================================
 * Scene (Engine3D/Scene.cs) as filled by Renderer.RaytraceGeometry (Renderer.cs:1513-1528)
 * ========================================================================================= */
struct SceneLight {
    double ambientLight_intensity, specularLight_shininess;
    Vec directionalLightDir_Model, directionalLightDir_View;
    Vec positionalLightPos_Model, positionalLightPos_View;
    bool pointLighting, specularLighting;
};

}  // namespace

/* =========================================================================================
 * orc_scene: geometry_simple + geometry_subdivided + ExtraGeometryToRaytrace
 * ========================================================================================= */
struct orc_scene {
    std::vector<Triangle> tris;          /* geometry_simple, Renderer.cs:1452-1469 */
    AxisAlignedBox box;                  /* AxisAlignedBox(model.Min, model.Max), :1487 */
    bool haveBox = false;
    SpatialSubdivision tree;             /* geometry_subdivided */
    bool haveTree = false;
    std::vector<orc_prim> extra;         /* ExtraGeometryToRaytrace in insertion order */
    std::vector<Sphere> spheres;         /* parallel storage, indexed through extraIdx */
    std::vector<Plane> planes;
    std::vector<Triangle> extraTris;
    std::vector<AxisAlignedBox> extraBoxes;
    std::vector<int> extraIdx;
    std::vector<uint8_t> staticCache;    /* ShadowMethod's Texture3DCache<byte>(128): empty until the first static frame */
};

namespace {

/* GeometryCollection.IntersectRay over all model triangles (GeometryCollection.cs:44-69) */
bool BruteIntersect(const orc_scene& s, Vec start, Vec dir, Hit& out, Counters& c) {
    Hit closest;
    closest.rayFrac = DBL_MAX;
    for (const Triangle& tri : s.tris) {
        Hit h;
        if (tri.IntersectRay(start, dir, h) && h.rayFrac < closest.rayFrac) closest = h;
        c.geomTests++;
    }
    if (closest.rayFrac == DBL_MAX) return false;
    out = closest;
    return true;
}

/* ORC_MODE_NEAREST: the tree's clip + offset (SpatialSubdivision.cs:381-419), then the global
 * nearest hit among hits inside the ROOT box (every leaf box is inside it, :652), first-listed
 * (= lowest index) wins ties.  This is what a correct BVH must return; it equals the reference
 * tree's answer except in boundary cases closer than 1e-10 to a leaf-box face. */
bool NearestIntersect(const orc_scene& s, Vec start, Vec dir, Hit& out, Counters& c) {
    Vec end = start + dir * 10000;
    Vec originalStart = start;
    if (!s.box.ClipLineSegment(start, end)) return false;
    double rayFracOffset = Distance(originalStart, start) / Length(dir);
    Hit closest;
    closest.rayFrac = DBL_MAX;
    for (const Triangle& tri : s.tris) {
        Hit h;
        if (tri.IntersectRay(start, dir, h) && h.rayFrac < closest.rayFrac) {
            if (s.box.ContainsPoint(h.pos)) closest = h;
        }
        c.geomTests++;
    }
    if (closest.rayFrac == DBL_MAX) return false;
    out = closest;
    out.rayFrac += rayFracOffset;
    return true;
}

bool ModelIntersect(const orc_scene& s, int mode, Vec start, Vec dir, Hit& out, Counters& c) {
    switch (mode) {
        case ORC_MODE_REF_TREE: return s.tree.IntersectRay(start, dir, out, c);
        case ORC_MODE_BRUTE: return BruteIntersect(s, start, dir, out, c);
        default: return NearestIntersect(s, start, dir, out, c);
    }
}

/* The root geometry below the decorators (Renderer.cs:1536-1549): the model, or -- when extra
 * geometry exists -- a GeometryCollection of [extra..., model] (GeometryCollection.cs:44-69). */
bool RootIntersect(const orc_scene& s, int mode, Vec start, Vec dir, Hit& out, Counters& c) {
    if (s.extra.empty()) return ModelIntersect(s, mode, start, dir, out, c);
    Hit closest;
    closest.rayFrac = DBL_MAX;
    for (size_t i = 0; i < s.extra.size(); i++) {
        Hit h;
        bool ok = false;
        switch (s.extra[i].kind) {
            case 0: ok = s.spheres[s.extraIdx[i]].IntersectRay(start, dir, h); break;
            case 1: ok = s.planes[s.extraIdx[i]].IntersectRay(start, dir, h); break;
            case 4: ok = s.extraBoxes[s.extraIdx[i]].IntersectRay(start, dir, h); c.geomTests += 5; break;   /* NumRayTests = 6 planes, AxisAlignedBox.cs:70 */
            default: ok = s.extraTris[s.extraIdx[i]].IntersectRay(start, dir, h); break;
        }
        if (ok && h.rayFrac < closest.rayFrac) closest = h;
        c.geomTests++;
    }
    {
        Hit h;
        if (ModelIntersect(s, mode, start, dir, h, c) && h.rayFrac < closest.rayFrac) closest = h;
    }
    if (closest.rayFrac == DBL_MAX) return false;
    out = closest;
    return true;
}

/* ShadingMethod.CalcLighting + CalcLightingIntensity (ShadingMethod.cs:93-177), white materials */
double CalcLightingIntensity(const SceneLight& scene, Vec point, Vec normal) {
    Vec dirToLight;
    if (scene.pointLighting) {
        dirToLight = scene.positionalLightPos_View - point;
        Normalise(dirToLight);
    } else {
        dirToLight = -scene.directionalLightDir_View;
    }
    double diffuseIntensity = Dot(dirToLight, normal);
    diffuseIntensity = (0.0 > diffuseIntensity) ? 0.0 : diffuseIntensity;      /* Math.Max(0.0, x): NaN and -0.0 pass through */
    double specularIntensity = 0.0;
    if (scene.specularLighting) {
        Vec dirToCamera = -point;
        Normalise(dirToCamera);
        Vec reflectedLightDir = 2.0 * Dot(dirToLight, normal) * normal - dirToLight;
        double cosOfAngle = Dot(reflectedLightDir, dirToCamera);
        specularIntensity = std::pow(cosOfAngle, scene.specularLight_shininess);
        specularIntensity = (0.0 > specularIntensity) ? 0.0 : specularIntensity;
    }
    /* Color * double with material (1,1,1): 1.0*x == x exactly; per-channel sum, all channels equal */
    double ch = 1.0 * scene.ambientLight_intensity + 1.0 * diffuseIntensity + 1.0 * specularIntensity;
    ch = (ch < 1.0) ? ch : (std::isnan(ch) ? ch : 1.0);         /* Math.Min(color.r, 1.0) */
    return ch;                                                   /* Math.Max(Math.Max(r,g),b), r==g==b */
}

struct FrameCtx {
    const orc_scene* scene;
    const orc_frame* f;
    InstanceXf xf;
    SceneLight light;
    std::vector<Vec> areaLightOffsets;
    int shadowSamples;
    int mode;
    /* static soft-shadow cache (ShadowMethod.cs:75-83, Texture3DCache.cs:95-135), 128^3 bytes, 0 = empty; owned by
     * the scene (it outlives a frame, like the reference's ShadowMethod), filled in the order orc_render defines */
    std::vector<uint8_t>* staticCache;
};

/* ShadowMethod.TraceRaysForSoftShadows (ShadowMethod.cs:144-179).  The reference traces shadow rays
 * through the shading decorator as well (Renderer.cs:1625); its result is discarded, so skipped. */
double TraceRaysForSoftShadows(const FrameCtx& fc, Vec surfacePos, Vec surfaceNormal, Counters& c) {
    const double shadowProbeOffset = 0.001;
    int rayEscapeCount = 0;
    for (int i = 0; i < fc.shadowSamples; i++) {
        Vec dirLightToSurface, shadowRayStart;
        Vec shadowRayEnd = surfacePos + surfaceNormal * shadowProbeOffset;
        if (fc.light.pointLighting) {
            Vec lightSource = fc.light.positionalLightPos_Model + fc.areaLightOffsets[i];
            dirLightToSurface = shadowRayEnd - lightSource;
            shadowRayStart = lightSource;
        } else {
            dirLightToSurface = fc.light.directionalLightDir_Model;
            shadowRayStart = shadowRayEnd + dirLightToSurface * 1000.0 + fc.areaLightOffsets[i];
        }
        Hit sh;
        bool hit = RootIntersect(*fc.scene, fc.mode, shadowRayStart, dirLightToSurface, sh, c);
        if (!hit || sh.rayFrac > 1.0) rayEscapeCount++;
    }
    return (double)rayEscapeCount / (double)fc.shadowSamples;
}

/* The decorator chain for one camera ray, flags as set per frame (Renderer.cs:1590-1649):
 * LightFieldColor(off) > AmbientOcclusion(off) > Shadow > PathTracing(off) > Shading > LightFieldTri(off) > root
 * then TraceRayComplex (Renderer.cs:1850-1885): miss -> BackgroundColorWithAlpha.
 *
 * max_bounces > 0 is the config-5 EXTENSION (no counterpart in the reference; "parity unpinned", the definition is
 * this code): a Whitted mirror bounce.  After the chain has coloured a hit, the ray is reflected about the surface
 * normal, r = dir - n * (2.0 * dir.n), restarted at pos + n * 0.001 (the raySurfaceOffset of PathTracingMethod.cs:10),
 * traced through the same chain, and the two packed colours are blended per channel with integer arithmetic:
 * c = ((surface * (255 - k)) >> 8) + ((reflected * k) >> 8), k = (byte)(reflectivity * 255), alpha 0xFF.  A reflected
 * ray that misses sees the background colour.  Depth is limited to max_bounces reflections. */
uint32_t ChainColor(const FrameCtx& fc, const Hit& info0, Counters& secondary) {
    Hit info = info0;
    if (fc.f->flags & ORC_F_SHADING) {                          /* ShadingMethod.IntersectRay :36-68 */
        Vec pos_View = fc.xf.TransformPosToView(info.pos);
        Vec normal_View = fc.xf.TransformDirection(info.normal);
        double intensity = CalcLightingIntensity(fc.light, pos_View, normal_View);
        uint8_t lightIntensityByte = ToByte(255 * intensity);
        info.color = ModulatePackedColor(info.color, lightIntensityByte);
    }
    if ((fc.f->flags & ORC_F_SHADOWS) && (fc.f->flags & ORC_F_STATIC_SHADOWS)) {
        const int n = 128;                                      /* staticShadowRes, Renderer.cs:114 */
        auto cell = [&](double v) { int k = (int)((v + 0.5) * (n - 1)); return std::min(std::max(k, 0), n - 1); };
        size_t idx = (size_t)cell(info.pos.x) * n * n + (size_t)cell(info.pos.y) * n + (size_t)cell(info.pos.z);   /* Texture3DCache.cs:102-104 */
        uint8_t sample = (*fc.staticCache)[idx];
        if (sample == 0) {
            double v = TraceRaysForSoftShadows(fc, info.pos, info.normal, secondary) * 254 + 1;   /* ShadowMethod.cs:80 */
            sample = (uint8_t)(int)v;
            if (sample == 0) sample = 1;
            (*fc.staticCache)[idx] = sample;
        }
        info.color = ModulatePackedColor(info.color, sample);
    } else if (fc.f->flags & ORC_F_SHADOWS) {                   /* ShadowMethod.IntersectRay :93-121 */
        uint8_t lightIntensityByte = ToByte(TraceRaysForSoftShadows(fc, info.pos, info.normal, secondary) * 255);
        info.color = ModulatePackedColor(info.color, lightIntensityByte);
    }
    return info.color;
}

uint32_t TraceRayComplex(const FrameCtx& fc, Vec start, Vec dir, Counters& primary, Counters& secondary) {
    const uint32_t background = fc.f->background_argb | 0xFF000000u;   /* Renderer.cs:325-331,1860 */
    Hit info;
    if (!RootIntersect(*fc.scene, fc.mode, start, dir, info, primary)) return background;
    const int maxBounces = fc.f->max_bounces;
    if (maxBounces <= 0) return ChainColor(fc, info, secondary);

    /* ---- extension: mirror bounces ---- */
    uint32_t surface[17];
    int levels = 0;
    uint32_t tail = background;                                  /* what the deepest ray saw */
    bool tailIsSurface = false;
    for (;;) {
        surface[levels++] = ChainColor(fc, info, secondary);
        if (levels > maxBounces) { tailIsSurface = true; break; }
        Vec n = info.normal;
        Vec r = dir - n * (2.0 * Dot(dir, n));
        Vec rstart = info.pos + n * 0.001;
        Hit next;
        if (!RootIntersect(*fc.scene, fc.mode, rstart, r, next, secondary)) break;
        start = rstart; dir = r; info = next;
    }
    uint32_t k = ToByte(fc.f->reflectivity * 255.0);
    uint32_t color;
    int i;
    if (tailIsSurface) { color = surface[levels - 1]; i = levels - 2; }
    else { color = tail; i = levels - 1; }
    for (; i >= 0; --i) {
        uint32_t sfc = surface[i];
        uint32_t out = 255u << 24;
        for (int sh = 16; sh >= 0; sh -= 8) {
            uint32_t a = (sfc >> sh) & 0xffu, b = (color >> sh) & 0xffu;
            uint32_t c = ((a * (255u - k)) >> 8) + ((b * k) >> 8);
            out |= (c & 0xffu) << sh;
        }
        color = out;
    }
    return color;
}

/* Renderer.RaytraceBlock, one pixel (Renderer.cs:1718-1828) */
uint32_t RenderPixel(const FrameCtx& fc, int col, int row, Vec start_World, Counters& primary, Counters& secondary, int64_t& rays) {
    const orc_frame& f = *fc.f;
    const int width = f.width, height = f.height;
    const double aspectRatio = (double)height / (double)width;   /* Renderer.cs:621 */
    const double fieldOfViewDepth = fc.xf.fovDepth;
    const int n = f.sub_pixel_res;
    if (n == 1) {
        Vec dir_View = V(-((double)col / width - 0.5), -((double)row / height - 0.5) * aspectRatio, fieldOfViewDepth);
        Vec dir_World = fc.xf.TransformDirectionReverse(dir_View);
        rays++;
        return TraceRayComplex(fc, start_World, dir_World, primary, secondary);
    }
    const bool focalBlur = (f.flags & ORC_F_FOCAL_BLUR) != 0;
    int sumR = 0, sumG = 0, sumB = 0;
    Vec pixelFocalPt_World = V(0, 0, 0);
    if (focalBlur) {
        Vec dir_View = V(-((double)col / width - 0.5), -((double)row / height - 0.5) * aspectRatio, fieldOfViewDepth);
        Vec dir_World = fc.xf.TransformDirectionReverse(dir_View);
        pixelFocalPt_World = dir_World * f.focal_depth + start_World;
    }
    for (int subX = 0; subX < n; subX++) {
        for (int subY = 0; subY < n; subY++) {
            double fracSubX = (double)subX / (n - 1) - 0.5;
            double fracSubY = (double)subY / (n - 1) - 0.5;
            Vec subStart_World;
            if (focalBlur) {
                Vec subStart_View = V(fracSubX / width * f.focal_blur_strength,
                                      fracSubY / height * f.focal_blur_strength,
                                      -fc.xf.positionZ);
                subStart_World = fc.xf.TransformDirectionReverse(subStart_View);
            } else {
                subStart_World = start_World;
            }
            Vec dir_World;
            if (focalBlur) {
                dir_World = pixelFocalPt_World - subStart_World;
            } else {
                Vec dir_View = V(-((col + fracSubX) / width - 0.5),
                                 -((row + fracSubY) / height - 0.5) * aspectRatio,
                                 fieldOfViewDepth);
                dir_World = fc.xf.TransformDirectionReverse(dir_View);
            }
            rays++;
            uint32_t color = TraceRayComplex(fc, subStart_World, dir_World, primary, secondary);
            sumR += (uint8_t)((color >> 16) & 0xff);            /* Surface.UnpackRgb, Surface.cs:105-110 */
            sumG += (uint8_t)((color >> 8) & 0xff);
            sumB += (uint8_t)(color & 0xff);
        }
    }
    sumR /= n * n;
    sumG /= n * n;
    sumB /= n * n;
    return (255u << 24) + ((uint32_t)(uint8_t)sumR << 16) + ((uint32_t)(uint8_t)sumG << 8) + (uint8_t)sumB; /* PackRgb */
}

void FillAreaLightOffsets(int32_t seed, int count, std::vector<Vec>& out) {   /* ShadowMethod.cs:63-73 */
    DotNetRandom random(seed);
    out.resize(count);
    for (int i = 0; i < count; i++) {
        double x = random.NextDouble() * 2 - 1;
        double y = random.NextDouble() * 2 - 1;
        double z = random.NextDouble() * 2 - 1;
        Vec offset = V(x, y, z);
        Normalise(offset);
        offset = offset * 0.2;                                               /* offset *= 0.2 */
        out[i] = offset;
    }
}

bool RowOwned(const orc_frame& f, int row) {
    if (f.strip_count <= 0) return true;
    return ((row / f.strip_rows) % f.strip_count) == f.strip_index;
}

}  // namespace

/* =========================================================================================
 * C interface
 * ========================================================================================= */
extern "C" {

struct orc_random { DotNetRandom r; explicit orc_random(int32_t s) : r(s) {} };
orc_random* orc_random_new(int32_t seed) { return new orc_random(seed); }
void orc_random_free(orc_random* r) { delete r; }
int32_t orc_random_next(orc_random* r) { return r->r.Next(); }
int32_t orc_random_next_max(orc_random* r, int32_t max) { return r->r.Next(max); }
double orc_random_next_double(orc_random* r) { return r->r.NextDouble(); }
void orc_random_next_ints(orc_random* r, int64_t n, int32_t* out) { for (int64_t i = 0; i < n; i++) out[i] = r->r.Next(); }
void orc_random_next_doubles(orc_random* r, int64_t n, double* out) { for (int64_t i = 0; i < n; i++) out[i] = r->r.NextDouble(); }

orc_scene* orc_scene_new(void) { return new orc_scene(); }
void orc_scene_free(orc_scene* s) { delete s; }

void orc_scene_reset_shadow_cache(orc_scene* s) { if (s) s->staticCache.clear(); }

int orc_scene_set_triangles(orc_scene* s, const double* v9, const uint32_t* argb, int64_t n,
                            const double box_min[3], const double box_max[3]) {
    s->staticCache.clear();                                     /* new model: new renderer state */
    s->tris.clear();
    s->tris.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const double* p = v9 + 9 * i;
        Triangle t(V(p[0], p[1], p[2]), V(p[3], p[4], p[5]), V(p[6], p[7], p[8]), argb[i]);
        t.triangleIndex = (int)i;                                /* Renderer.cs:1465 */
        s->tris.push_back(t);
    }
    s->box = AxisAlignedBox(V(box_min[0], box_min[1], box_min[2]), V(box_max[0], box_max[1], box_max[2]));
    s->haveBox = true;
    s->haveTree = false;
    return 0;
}

int orc_scene_set_extra(orc_scene* s, const orc_prim* prims, int32_t n) {
    s->extra.assign(prims, prims + n);
    s->spheres.clear(); s->planes.clear(); s->extraTris.clear(); s->extraBoxes.clear(); s->extraIdx.clear();
    for (int i = 0; i < n; i++) {
        const orc_prim& p = prims[i];
        switch (p.kind) {
            case 0:
                s->extraIdx.push_back((int)s->spheres.size());
                s->spheres.emplace_back(V(p.p[0], p.p[1], p.p[2]), p.p[3], p.argb);
                break;
            case 1: {
                s->extraIdx.push_back((int)s->planes.size());
                Plane pl(V(p.p[0], p.p[1], p.p[2]), V(p.p[3], p.p[4], p.p[5]));
                pl.color = p.argb;
                s->planes.push_back(pl);
                break;
            }
            case 2:
                s->extraIdx.push_back((int)s->extraTris.size());
                s->extraTris.emplace_back(V(p.p[0], p.p[1], p.p[2]), V(p.p[3], p.p[4], p.p[5]), V(p.p[6], p.p[7], p.p[8]), p.argb);
                break;
            case 4:                                              /* AxisAlignedBox(min, max): its planes are Color.White (Plane.cs:28) */
                if (!(p.p[0] < p.p[3] && p.p[1] < p.p[4] && p.p[2] < p.p[5])) return -1;   /* Contract.Requires, AxisAlignedBox.cs:17-19 */
                s->extraIdx.push_back((int)s->extraBoxes.size());
                s->extraBoxes.emplace_back(V(p.p[0], p.p[1], p.p[2]), V(p.p[3], p.p[4], p.p[5]));
                break;
            default: return -1;
        }
    }
    return 0;
}

int orc_scene_build_tree(orc_scene* s, int32_t max_depth, int32_t max_per_leaf) {
    if (!s->haveBox) return -1;
    if (max_depth <= 0) max_depth = 15;                          /* SpatialSubdivision.cs:269-270 */
    if (max_per_leaf <= 0) max_per_leaf = 25;
    s->tree = SpatialSubdivision();
    if (!s->tree.Build(s->tris, s->box, max_depth, max_per_leaf)) return -2;
    s->haveTree = true;
    return 0;
}

void orc_scene_tree_stats(const orc_scene* s, int32_t out[4]) {
    out[0] = s->tree.TreeDepth; out[1] = s->tree.NumNodes; out[2] = s->tree.NumLeafNodes; out[3] = s->tree.NumInternalNodes;
}

void orc_instance_matrices(const double position[3], double yaw, double pitch, double roll,
                           double transform[12], double inv_transform[12]) {
    Vec P = V(position[0], position[1], position[2]);
    /* Instance.cs:134-135 */
    Matrix t = MakeTranslationMatrix(P) * MakeRollMatrix(roll) * MakePitchMatrix(pitch) * MakeYawMatrix(yaw);
    Matrix it = MakeYawMatrix(-yaw) * MakePitchMatrix(-pitch) * MakeRollMatrix(-roll) * MakeTranslationMatrix(-P);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) { transform[r * 4 + c] = t.m[r][c]; inv_transform[r * 4 + c] = it.m[r][c]; }
}

double orc_default_fov_depth(void) {
    /* Renderer.cs:97-101 : constants folded in the same order */
    const double fieldOfViewDeg = 45.0;
    const double fieldOfViewRad = fieldOfViewDeg / 180.0 * M_PI;
    return 0.5 / std::tan(fieldOfViewRad / 2);
}

void orc_area_light_offsets(int32_t seed, int32_t count, double* out3) {
    std::vector<Vec> v;
    FillAreaLightOffsets(seed, count, v);
    for (int i = 0; i < count; i++) { out3[3 * i] = v[i].x; out3[3 * i + 1] = v[i].y; out3[3 * i + 2] = v[i].z; }
}

/* ShadingMethod.IntersectRay's colour step (ShadingMethod.cs:36-68) for recorded intersections: pure arithmetic, no
 * traversal -- what the GPU-vs-CPU census of Math.Pow (ocml vs glibc) compares over >= 1e7 surface points */
int orc_shade_points(const orc_frame* f, int64_t n, const double* pos, const double* normal, const uint32_t* color, uint32_t* out, int32_t threads) {
    InstanceXf xf;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) { xf.t[r][c] = f->transform[r * 4 + c]; xf.it[r][c] = f->inv_transform[r * 4 + c]; }
    xf.positionZ = f->position_z;
    xf.fovDepth = f->fov_depth;
    SceneLight light;
    light.directionalLightDir_View = V(f->light_dir_view[0], f->light_dir_view[1], f->light_dir_view[2]);
    light.positionalLightPos_View = V(f->light_pos_view[0], f->light_pos_view[1], f->light_pos_view[2]);
    light.directionalLightDir_Model = xf.TransformDirectionReverse(light.directionalLightDir_View);
    light.positionalLightPos_Model = xf.TransformPosFromView(light.positionalLightPos_View);
    light.pointLighting = (f->flags & ORC_F_POINT_LIGHT) != 0;
    light.specularLighting = (f->flags & ORC_F_SPECULAR) != 0;
    light.ambientLight_intensity = f->ambient;
    light.specularLight_shininess = f->shininess;
    int nt = std::max(1, (int)threads);
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++) {
        pool.emplace_back([&, t]() {
            for (int64_t i = (n * t) / nt; i < (n * (t + 1)) / nt; i++) {
                Vec pos_View = xf.TransformPosToView(V(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
                Vec normal_View = xf.TransformDirection(V(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]));
                double intensity = CalcLightingIntensity(light, pos_View, normal_View);
                out[i] = ModulatePackedColor(color[i], ToByte(255 * intensity));
            }
        });
    }
    for (auto& th : pool) th.join();
    return 0;
}

static int RenderColumns(const orc_scene* s, const orc_frame* f, int32_t* pixels, uint64_t stats[4], int32_t threads, int colBegin, int colEnd);

int orc_render(const orc_scene* s, const orc_frame* f, int32_t* pixels, uint64_t stats[4], int32_t threads) {
    return RenderColumns(s, f, pixels, stats, threads, 0, f->width);
}

/* the same frame restricted to columns [col_begin, col_end): every pixel it draws is the pixel orc_render draws there (pixels
 * are independent, Renderer.cs:1690: RaytraceBlock takes left / sizeX); used for the CPU baseline's centred crop only */
int orc_render_window(const orc_scene* s, const orc_frame* f, int32_t* pixels, uint64_t stats[4], int32_t threads, int32_t col_begin, int32_t col_end) {
    if (col_begin < 0 || col_end > f->width || col_begin >= col_end) return -8;
    if (f->flags & ORC_F_STATIC_SHADOWS) return -7;              /* the static cache's fill order is defined over whole rows */
    return RenderColumns(s, f, pixels, stats, threads, col_begin, col_end);
}

static int RenderColumns(const orc_scene* s, const orc_frame* f, int32_t* pixels, uint64_t stats[4], int32_t threads, int colBegin, int colEnd) {
    if (!s->haveBox || s->tris.empty()) return -3;               /* no model pinned: Renderer.cs:736-739 */
    if (f->trace_mode == ORC_MODE_REF_TREE && !s->haveTree) return -4;
    if (f->sub_pixel_res < 1) return -5;
    if (f->max_bounces < 0 || f->max_bounces > 16 || !(f->reflectivity >= 0.0 && f->reflectivity <= 1.0)) return -6;
    FrameCtx fc;
    fc.scene = s;
    fc.f = f;
    fc.mode = f->trace_mode;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) { fc.xf.t[r][c] = f->transform[r * 4 + c]; fc.xf.it[r][c] = f->inv_transform[r * 4 + c]; }
    fc.xf.positionZ = f->position_z;
    fc.xf.fovDepth = f->fov_depth;
    /* Renderer.cs:1513-1528 */
    Vec lightDirView = V(f->light_dir_view[0], f->light_dir_view[1], f->light_dir_view[2]);
    Vec lightPosView = V(f->light_pos_view[0], f->light_pos_view[1], f->light_pos_view[2]);
    fc.light.directionalLightDir_View = lightDirView;
    fc.light.positionalLightPos_View = lightPosView;
    fc.light.directionalLightDir_Model = fc.xf.TransformDirectionReverse(lightDirView);
    fc.light.positionalLightPos_Model = fc.xf.TransformPosFromView(lightPosView);
    fc.light.pointLighting = (f->flags & ORC_F_POINT_LIGHT) != 0;
    fc.light.specularLighting = (f->flags & ORC_F_SPECULAR) != 0;
    fc.light.ambientLight_intensity = f->ambient;
    fc.light.specularLight_shininess = f->shininess;
    fc.shadowSamples = f->shadow_samples > 0 ? f->shadow_samples : 100;   /* ShadowMethod.cs:9 */
    if (f->area_light_offsets) {
        fc.areaLightOffsets.resize(fc.shadowSamples);
        for (int i = 0; i < fc.shadowSamples; i++)
            fc.areaLightOffsets[i] = V(f->area_light_offsets[3 * i], f->area_light_offsets[3 * i + 1], f->area_light_offsets[3 * i + 2]);
    } else {
        FillAreaLightOffsets(f->random_seed, fc.shadowSamples, fc.areaLightOffsets);
    }

    /* Renderer.cs:1652-1653 */
    int startRow = std::min(std::max(0, f->start_row), f->height - 1);
    int endRow = std::min(std::max(0, f->end_row), f->height - 1);

    /* rows this call owns, in order; compact output position when strips are on */
    std::vector<int> rows;
    for (int r = startRow; r <= endRow; r++)
        if (RowOwned(*f, r)) rows.push_back(r);

    Vec start_World = fc.xf.TransformDirectionReverse(V(0, 0, -fc.xf.positionZ));   /* Renderer.cs:1717 */

    int nthreads = threads > 0 ? threads : 1;
    if (f->flags & ORC_F_STATIC_SHADOWS) {
        if (f->max_bounces > 0 || f->strip_count > 1) return -7;    /* not defined for the bounce extension / a split frame */
        orc_scene* ms = const_cast<orc_scene*>(s);                  /* the cache is renderer state */
        if (ms->staticCache.empty()) ms->staticCache.assign((size_t)128 * 128 * 128, 0);
        fc.staticCache = &ms->staticCache;
        nthreads = 1;
    }
    /* work items = 64-pixel chunks of a row (the reference fans out row BLOCKS, Renderer.cs:1659-1670; pixels are
     * independent, so the partition does not change any pixel) */
    const int chunk = 64;
    const size_t chunksPerRow = (size_t)(colEnd - colBegin + chunk - 1) / chunk;
    std::atomic<size_t> nextItem{0};
    std::vector<Counters> prim(nthreads), sec(nthreads);
    std::vector<int64_t> rays(nthreads, 0);
    auto worker = [&](int tid) {
        for (;;) {
            size_t item = nextItem.fetch_add(1);
            if (item >= rows.size() * chunksPerRow) break;
            size_t i = item / chunksPerRow;
            int c0 = colBegin + (int)(item % chunksPerRow) * chunk, c1 = std::min(colEnd, c0 + chunk);
            int row = rows[i];
            int32_t* dst = (f->strip_count > 0) ? pixels + (size_t)i * f->width : pixels + (size_t)row * f->width;
            for (int col = c0; col < c1; col++)
                dst[col] = (int32_t)RenderPixel(fc, col, row, start_World, prim[tid], sec[tid], rays[tid]);
        }
    };
    if (f->flags & ORC_F_STATIC_SHADOWS) {
        /* rayTraceShadowsStatic.  The reference fills the cache from rayTraceConcurrency row-block tasks that race
         * (Renderer.cs:1659-1670, Texture3DCache.cs:117-135; its own test notes that the thread count changes pixels).
         * Deterministic definition used here, PINNED by the two goldens of RendererTests.RaytraceStaticShadow
         * (shading_staticShadows.bmp, noShading_staticShadows.bmp, 0 differing pixels; plain scan order differs in 9
         * pixels, all on the block seams): the tasks advance in lockstep -- row r of every block, blocks in ascending
         * order, before row r + 1; columns ascending inside a row -- and a cell keeps the value computed for its
         * first requester. */
        const int conc = f->concurrency > 0 ? f->concurrency : 4;   /* Renderer.cs:92 */
        const int numRows = (int)rows.size();
        const int blockHeight = (numRows - 1 + conc) / conc;        /* :1661 */
        for (int r = 0; r < blockHeight; r++)
            for (int b = 0; b * blockHeight < numRows; b++) {
                const int i = b * blockHeight + r;
                if (i >= numRows) continue;
                const int row = rows[i];
                int32_t* dst = (f->strip_count > 0) ? pixels + (size_t)i * f->width : pixels + (size_t)row * f->width;
                for (int col = 0; col < f->width; col++)
                    dst[col] = (int32_t)RenderPixel(fc, col, row, start_World, prim[0], sec[0], rays[0]);
            }
    } else if (nthreads == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(worker, t);
        for (auto& t : th) t.join();
    }
    if (stats) {
        stats[0] = stats[1] = stats[2] = stats[3] = 0;
        for (int t = 0; t < nthreads; t++) {
            stats[0] += (uint64_t)rays[t];
            stats[1] += (uint64_t)prim[t].geomTests;
            stats[2] += (uint64_t)prim[t].nodeVisits;
            stats[3] += (uint64_t)prim[t].leafVisits;
        }
    }
    return 0;
}

int orc_trace(const orc_scene* s, int32_t target, int64_t n, const double* starts, const double* dirs,
              uint8_t* hit, double* ray_frac, double* pos, double* normal, uint32_t* color,
              int32_t* tri_index, int32_t* counters) {
    if (target == 1 && !s->haveTree) return -4;
    for (int64_t i = 0; i < n; i++) {
        Vec st = V(starts[3 * i], starts[3 * i + 1], starts[3 * i + 2]);
        Vec d = V(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        Hit h;
        Counters c;
        bool ok;
        switch (target) {
            case 0: ok = BruteIntersect(*s, st, d, h, c); break;
            case 1: ok = s->tree.IntersectRay(st, d, h, c); break;
            case 2: ok = RootIntersect(*s, s->haveTree ? ORC_MODE_REF_TREE : ORC_MODE_BRUTE, st, d, h, c); break;
            case 3: ok = NearestIntersect(*s, st, d, h, c); break;
            default: return -1;
        }
        if (hit) hit[i] = ok ? 1 : 0;
        if (ray_frac) ray_frac[i] = ok ? h.rayFrac : 0.0;
        if (pos) { pos[3 * i] = ok ? h.pos.x : 0; pos[3 * i + 1] = ok ? h.pos.y : 0; pos[3 * i + 2] = ok ? h.pos.z : 0; }
        if (normal) { normal[3 * i] = ok ? h.normal.x : 0; normal[3 * i + 1] = ok ? h.normal.y : 0; normal[3 * i + 2] = ok ? h.normal.z : 0; }
        if (color) color[i] = ok ? h.color : 0;
        if (tri_index) tri_index[i] = ok ? h.triIndex : -1;
        if (counters) { counters[3 * i] = (int32_t)c.geomTests; counters[3 * i + 1] = (int32_t)c.nodeVisits; counters[3 * i + 2] = (int32_t)c.leafVisits; }
    }
    return 0;
}

/* =========================================================================================
 * 3DS loader + Model post-processing
 *   SalmonViewer.ThreeDSFile (Engine3D/3dsLoader/ThreeDSFile.cs:132-662)
 *   Model.Load3ds (Engine3D/Model.cs:522-653), Model.PostProcessGeometry (Model.cs:750-831)
 * ========================================================================================= */
struct orc_model {
    std::vector<Vec> vertices;
    std::vector<int> tri;        /* 3 vertex indices per triangle */
    std::vector<uint32_t> argb;  /* per triangle: PackColorAndAlpha(diffuse, 1.0), Renderer.cs:1463 */
    Vec min, max;
};

namespace {

struct Material3ds { float diffuse[3] = {0.0f, 0.0f, 0.0f}; };   /* Material.cs:32-35 defaults */

struct Entity3ds {
    std::vector<Vec> vertices;          /* (x, z, -y) swap already applied, ThreeDSFile.cs:627 */
    std::vector<int> tris;              /* 3 per face */
    std::vector<int> faceMaterial;      /* index into materials, -1 = Triangle.defaultMaterial */
    bool hasVerts = false, hasTris = false;
};

struct Reader3ds {
    const uint8_t* d; size_t len; size_t pos = 0; bool bad = false;
    uint16_t u16() { if (pos + 2 > len) { bad = true; return 0; } uint16_t v; std::memcpy(&v, d + pos, 2); pos += 2; return v; }
    uint32_t u32() { if (pos + 4 > len) { bad = true; return 0; } uint32_t v; std::memcpy(&v, d + pos, 4); pos += 4; return v; }
    float f32() { if (pos + 4 > len) { bad = true; return 0; } float v; std::memcpy(&v, d + pos, 4); pos += 4; return v; }
    uint8_t u8() { if (pos + 1 > len) { bad = true; return 0; } return d[pos++]; }
    std::string cstr() { std::string s; uint8_t b = u8(); while (b != 0 && !bad) { s.push_back((char)b); b = u8(); } return s; }
};

enum {
    C_COLOR_F = 0x0010, C_COLOR_24 = 0x0011, C_PRIMARY = 0x4D4D, C_OBJECTINFO = 0x3D3D, C_VERSION = 0x0002,
    C_MATERIAL = 0xAFFF, C_MATNAME = 0xA000, C_MATDIFFUSE = 0xA020, C_OBJECT = 0x4000, C_OBJECT_MESH = 0x4100,
    C_OBJECT_VERTICES = 0x4110, C_OBJECT_FACES = 0x4120, C_OBJECT_MATERIAL = 0x4130
};

struct Loader3ds {
    Reader3ds rd;
    std::vector<Material3ds> materials;
    std::map<std::string, int> materialByName;
    std::vector<Entity3ds> entities;

    /* ProcessColorChunk (ThreeDSFile.cs:423-452): only the FIRST colour sub-chunk is read */
    void ProcessColorChunk(float out[3]) {
        size_t childStart = rd.pos;
        uint16_t id = rd.u16(); uint32_t clen = rd.u32();
        float red = 1.0f, green = 1.0f, blue = 1.0f;
        if (id == C_COLOR_F) { red = rd.f32(); green = rd.f32(); blue = rd.f32(); }
        else if (id == C_COLOR_24) {
            red = (float)rd.u8() / 255.0f; green = (float)rd.u8() / 255.0f; blue = (float)rd.u8() / 255.0f;
        }
        rd.pos = childStart + clen;
        out[0] = red; out[1] = green; out[2] = blue;
    }
    /* ProcessMaterialChunk (:249-316) */
    void ProcessMaterialChunk(size_t end) {
        std::string name;
        Material3ds m;
        while (rd.pos < end && !rd.bad) {
            size_t cs = rd.pos;
            uint16_t id = rd.u16(); uint32_t clen = rd.u32();
            if (clen < 6) { rd.bad = true; break; }
            if (id == C_MATNAME) name = rd.cstr();
            else if (id == C_MATDIFFUSE) ProcessColorChunk(m.diffuse);
            rd.pos = cs + clen;
        }
        if (!materialByName.count(name)) {                       /* no duplicate names (:311-315) */
            materialByName[name] = (int)materials.size();
            materials.push_back(m);
        }
    }
    /* ProcessFaceChunk (:512-566) */
    void ProcessFaceChunk(size_t end, Entity3ds& e) {
        while (rd.pos < end && !rd.bad) {
            size_t cs = rd.pos;
            uint16_t id = rd.u16(); uint32_t clen = rd.u32();
            if (clen < 6) { rd.bad = true; break; }
            if (id == C_OBJECT_MATERIAL) {
                std::string materialName = rd.cstr();
                int mi = -1;
                auto it = materialByName.find(materialName);
                if (it != materialByName.end()) mi = it->second;
                int nfaces = rd.u16();
                for (int i = 0; i < nfaces; i++) {
                    int faceIndex = rd.u16();
                    if (faceIndex < (int)e.faceMaterial.size()) e.faceMaterial[faceIndex] = mi;
                    else rd.bad = true;                          /* C#: IndexOutOfRangeException */
                }
            }
            rd.pos = cs + clen;
        }
    }
    /* ProcessObjectChunk (:466-510) */
    void ProcessObjectChunk(size_t end, Entity3ds& e) {
        while (rd.pos < end && !rd.bad) {
            size_t cs = rd.pos;
            uint16_t id = rd.u16(); uint32_t clen = rd.u32();
            if (clen < 6) { rd.bad = true; break; }
            size_t ce = cs + clen;
            if (id == C_OBJECT_MESH) {
                ProcessObjectChunk(ce, e);
            } else if (id == C_OBJECT_VERTICES) {                /* ReadVertices :611-634 */
                int numVerts = rd.u16();
                e.vertices.resize(numVerts);
                for (int i = 0; i < numVerts; i++) {
                    float f1 = rd.f32(), f2 = rd.f32(), f3 = rd.f32();
                    e.vertices[i] = V((double)f1, (double)f3, (double)(-f2));
                }
                e.hasVerts = true;
            } else if (id == C_OBJECT_FACES) {                   /* ReadTriangles :636-657 */
                int numTris = rd.u16();
                e.tris.resize((size_t)numTris * 3);
                e.faceMaterial.assign(numTris, -1);
                for (int i = 0; i < numTris; i++) {
                    e.tris[3 * i] = rd.u16(); e.tris[3 * i + 1] = rd.u16(); e.tris[3 * i + 2] = rd.u16();
                    rd.u16();                                    /* flags */
                }
                e.hasTris = true;
                if (rd.pos < ce) ProcessFaceChunk(ce, e);
            }
            rd.pos = ce;
        }
    }
    /* ProcessChunk (:187-247) */
    void ProcessChunk(size_t end) {
        while (rd.pos < end && !rd.bad) {
            size_t cs = rd.pos;
            uint16_t id = rd.u16(); uint32_t clen = rd.u32();
            if (clen < 6) { rd.bad = true; break; }
            size_t ce = cs + clen;
            switch (id) {
                case C_VERSION: rd.u32(); break;                 /* not skipped to end (:241-244); len is 10 anyway */
                case C_OBJECTINFO: {
                    /* the first sub-chunk (mesh version) is read and skipped, then the rest is processed */
                    size_t os = rd.pos;
                    rd.u16(); uint32_t olen = rd.u32();
                    rd.pos = os + olen;
                    ProcessChunk(ce);
                    break;
                }
                case C_MATERIAL: ProcessMaterialChunk(ce); break;
                case C_OBJECT: {
                    rd.cstr();
                    Entity3ds e;
                    ProcessObjectChunk(ce, e);
                    if (e.hasVerts && e.hasTris) entities.push_back(std::move(e));
                    break;
                }
                default: break;
            }
            if (id != C_VERSION) rd.pos = ce;
        }
    }
};

}  // namespace

orc_model* orc_model_load_3ds(const uint8_t* data, size_t len, char* err, size_t errlen) {
    auto fail = [&](const char* msg) -> orc_model* { if (err && errlen) std::snprintf(err, errlen, "%s", msg); return nullptr; };
    Loader3ds L;
    L.rd.d = data; L.rd.len = len;
    uint16_t id = L.rd.u16(); uint32_t plen = L.rd.u32();
    if (L.rd.bad || id != C_PRIMARY) return fail("Not a proper 3DS file.");                 /* :166-169 */
    L.ProcessChunk(std::min<size_t>(plen, len));
    if (L.rd.bad) return fail("3DS file truncated or corrupt.");
    if (L.entities.empty()) return fail("No entities in model. 3DS file may be corrupt.");  /* Model.cs:553-556 */

    std::unique_ptr<orc_model> m(new orc_model());
    const double maxCoordinateSize = 1e6;                        /* Model.cs (maxCoordinateSize) */
    m->min = V(DBL_MAX, DBL_MAX, DBL_MAX);
    m->max = V(-DBL_MAX, -DBL_MAX, -DBL_MAX);                    /* double.MinValue */
    for (const Entity3ds& e : L.entities) {
        if (e.vertices.size() < 3) return fail("Entity has less than 3 vertices. 3DS file may be corrupt.");
        if (e.tris.empty()) return fail("Entity has no triangles. 3DS file may be corrupt.");
        int vertexOffset = (int)m->vertices.size();
        for (const Vec& v : e.vertices) {                        /* Model.cs:585-610 */
            double x = v.x, y = v.y, z = v.z;
            if (std::isnan(x) || std::isinf(x) || std::fabs(x) > maxCoordinateSize) x = 0.0;
            if (std::isnan(y) || std::isinf(y) || std::fabs(y) > maxCoordinateSize) y = 0.0;
            if (std::isnan(z) || std::isinf(z) || std::fabs(z) > maxCoordinateSize) z = 0.0;
            m->vertices.push_back(V(x, y, z));
            m->min = V(std::min(m->min.x, x), std::min(m->min.y, y), std::min(m->min.z, z));
            m->max = V(std::max(m->max.x, x), std::max(m->max.y, y), std::max(m->max.z, z));
        }
        size_t nt = e.tris.size() / 3;
        for (size_t i = 0; i < nt; i++) {                        /* Model.cs:620-643 */
            m->tri.push_back(vertexOffset + e.tris[3 * i]);
            m->tri.push_back(vertexOffset + e.tris[3 * i + 1]);
            m->tri.push_back(vertexOffset + e.tris[3 * i + 2]);
            Material3ds def;
            const Material3ds& mat = e.faceMaterial[i] >= 0 ? L.materials[e.faceMaterial[i]] : def;
            /* Model.cs:98-100 float -> double; Renderer.cs:1463 PackColorAndAlpha(diffuse, 1.0) */
            m->argb.push_back(PackColorAndAlpha((double)mat.diffuse[0], (double)mat.diffuse[1], (double)mat.diffuse[2], 1.0));
        }
    }
    for (int idx : m->tri)
        if (idx < 0 || idx >= (int)m->vertices.size()) return fail("Triangle vertex index out of range.");
    /* PostProcessGeometry (Model.cs:750-790) */
    Vec centre = V((m->min.x + m->max.x) / 2, (m->min.y + m->max.y) / 2, (m->min.z + m->max.z) / 2);
    Vec extent = V(m->max.x - m->min.x, m->max.y - m->min.y, m->max.z - m->min.z);
    double scaleFactor = 1.0 / std::max(std::max(extent.x, extent.y), extent.z);
    for (Vec& v : m->vertices) v = (v - centre) * scaleFactor;
    m->min = (m->min - centre) * scaleFactor;
    m->max = (m->max - centre) * scaleFactor;
    return m.release();
}

void orc_model_free(orc_model* m) { delete m; }
int64_t orc_model_num_triangles(const orc_model* m) { return (int64_t)(m->tri.size() / 3); }
int64_t orc_model_num_vertices(const orc_model* m) { return (int64_t)m->vertices.size(); }
void orc_model_get(const orc_model* m, double* v9, uint32_t* argb, double bmin[3], double bmax[3]) {
    size_t nt = m->tri.size() / 3;
    for (size_t i = 0; i < nt; i++) {
        for (int k = 0; k < 3; k++) {
            const Vec& v = m->vertices[m->tri[3 * i + k]];
            v9[9 * i + 3 * k] = v.x; v9[9 * i + 3 * k + 1] = v.y; v9[9 * i + 3 * k + 2] = v.z;
        }
        argb[i] = m->argb[i];
    }
    bmin[0] = m->min.x; bmin[1] = m->min.y; bmin[2] = m->min.z;
    bmax[0] = m->max.x; bmax[1] = m->max.y; bmax[2] = m->max.z;
}

}  /* extern "C" */
