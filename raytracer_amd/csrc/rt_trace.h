// Exact primitive tests, hit records and light set-up of the trace loop, shared by
// the brute-force and LBVH kernels.  Every function restates one piece of
// reference src/raytrace.cpp with the same operand types and evaluation order;
// shortcuts are taken only where they are provably equal to the reference's
// result (each one says why).
#pragma once

#include "rt_device.h"

#define RT_DMIN 0.001   // self-intersection window, a SQUARED distance (src/raytrace.cpp:139, :536)

__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }

// ---------------------------------------------------------------- spheres
// One iteration of RayTrace::intersectSpheres (src/raytrace.cpp:226-299) for one
// sphere: true when the sphere yields a candidate inside the window; d2 is the
// candidate's squared distance (a float widened to double in the reference), tf
// the float ray parameter that produced it.
// Window (src/raytrace.cpp:245,:258,:273,:286): t > 0 && d > dmin && d < dmax, strict.
__device__ __forceinline__ bool sphere_test(F3 o, F3 d, double dd, float4 s, float dmax, float& d2out,
                                   float& tfout) {
    F3 ec = o - f3(s);
    double dec = (double)dot3(d, ec);
    double det = (dec * dec) - dd * (double)(dot3(ec, ec) - (s.w * s.w));
    if (det < 0.0) return false;
    double sq = __builtin_sqrt(det);
    double t1 = (-dec + sq) / dd;
    double t2 = (-dec - sq) / dd;
    float tf1 = (float)t1, tf2 = (float)t2;
    float d1 = magsq3(d * tf1);
    float d2 = magsq3(d * tf2);
    // the root tried first is the smaller one by the reference's own test (:244)
    double ta, tb; float da, db, fa, fb;
    if (t1 < t2) { ta = t1; da = d1; fa = tf1; tb = t2; db = d2; fb = tf2; }
    else         { ta = t2; da = d2; fa = tf2; tb = t1; db = d1; fb = tf1; }
    if (ta > 0 && (double)da > RT_DMIN && da < dmax) { d2out = da; tfout = fa; return true; }
    if (tb > 0 && (double)db > RT_DMIN && db < dmax) { d2out = db; tfout = fb; return true; }
    return false;
}

// ---------------------------------------------------------------- triangles
struct TriHit {
    double beta, gamma;
    F3 point;
    float d2;
};

// One iteration of RayTrace::intersectTriangle (src/raytrace.cpp:314-334).
// a = v0, ab = v0 - v1, ac = v0 - v2 (precomputed with the same float
// subtractions the reference performs per test, :316-317).
//
// The reference forms tval, beta, gamma as DOUBLE quotients of float
// determinants and compares them with 0 and 1.  For finite, non-zero detA those
// comparisons depend only on signs and magnitudes of the float determinants (a
// correctly rounded quotient has the sign of its exact value and is <= 1 exactly
// when |num| <= |den|), so the three divisions are skipped for the triangles
// that fail them -- nearly all -- and performed, in double, only for survivors,
// where beta + gamma <= 1 needs the rounded quotients themselves.  Any
// non-finite or zero determinant takes the literal path.
// Window (src/raytrace.cpp:332): reject when dist < dmin || dist > dmax (inclusive).
__device__ __forceinline__ bool tri_test(F3 o, F3 d, F3 a, F3 ab, F3 ac, float dmax, TriHit& h) {
    F3 ae = a - o;
    float m1 = ac.y * d.z - d.y * ac.z;
    float m2 = ac.z * d.x - d.z * ac.x;
    float m3 = ac.x * d.y - ac.y * d.x;
    float detA = ab.x * m1 + ab.y * m2 + ab.z * m3;
    float n1 = ab.x * ae.y - ab.y * ae.x;
    float n2 = ae.x * ab.z - ab.x * ae.z;
    float n3 = ab.y * ae.z - ab.z * ae.y;
    float dett = ac.z * n1 + ac.y * n2 + ac.x * n3;
    float detb = ae.x * m1 + ae.y * m2 + ae.z * m3;
    float detg = d.z * n1 + d.y * n2 + d.x * n3;
    bool plain = (detA != 0.0f) && (__builtin_fabsf(detA) < __builtin_inff()) &&
                 (__builtin_fabsf(dett) < __builtin_inff()) &&
                 (__builtin_fabsf(detb) < __builtin_inff()) &&
                 (__builtin_fabsf(detg) < __builtin_inff());
    if (plain) {
        bool apos = detA > 0.0f;
        // tval = dett/detA > 0  -> behind the origin, skip (:321)
        if (dett != 0.0f && ((dett > 0.0f) == apos)) return false;
        // beta < 0 or beta > 1, gamma < 0 or gamma > 1 (:328)
        if (detb != 0.0f && ((detb > 0.0f) != apos)) return false;
        if (detg != 0.0f && ((detg > 0.0f) != apos)) return false;
        float aA = __builtin_fabsf(detA);
        if (__builtin_fabsf(detb) > aA || __builtin_fabsf(detg) > aA) return false;
    } else {
        double tval = (double)dett / (double)detA;
        if (tval > 0) return false;
    }
    double beta = (double)detb / (double)detA;
    double gamma = (double)detg / (double)detA;
    if (!(beta >= 0 && beta <= 1 && gamma >= 0 && gamma <= 1 && beta + gamma <= 1)) return false;
    // point = v0 + ((v1-v0)*beta + (v2-v0)*gamma) (:329-330); v1-v0 == -(v0-v1) exactly.
    F3 point = a + ((neg3(ab) * (float)beta) + (neg3(ac) * (float)gamma));
    float dist = magsq3(point - o);
    if ((double)dist < RT_DMIN || dist > dmax) return false;
    h.beta = beta; h.gamma = gamma; h.point = point; h.d2 = dist;
    return true;
}

// Shading normal of a triangle hit (src/raytrace.cpp:336-358): flat normals flip
// to face the ray; smooth normals flip per vertex, then blend as
// n0*beta + n1*gamma + n2*alpha (the reference's pairing) and normalise.
__device__ __forceinline__ F3 tri_normal(const float4* __restrict__ tn, int flag, F3 rd, const TriHit& h) {
    F3 n0 = f3(tn[0]);
    if (!flag) {
        double ang = (double)dot3(n0, rd);
        return (ang > 0) ? neg3(n0) : n0;
    }
    F3 n1 = f3(tn[1]), n2 = f3(tn[2]);
    double alpha = 1.0 - (h.beta + h.gamma);
    double ang1 = (double)dot3(n0, rd), ang2 = (double)dot3(n1, rd), ang3 = (double)dot3(n2, rd);
    F3 m1 = (ang1 > 0) ? neg3(n0) : n0;
    F3 m2 = (ang2 > 0) ? neg3(n1) : n1;
    F3 m3 = (ang3 > 0) ? neg3(n2) : n2;
    return norm3(((m1 * (float)h.beta) + (m2 * (float)h.gamma)) + (m3 * (float)alpha));
}

__device__ __forceinline__ void tri_load(const float4* __restrict__ t, F3& a, F3& ab, F3& ac) {
    float4 q0 = t[0], q1 = t[1], q2 = t[2];
    a = f3(q0.x, q0.y, q0.z);
    ab = f3(q0.w, q1.x, q1.y);
    ac = f3(q1.z, q1.w, q2.x);
}

// What a walk keeps of its winning candidate besides the primitive id and the squared
// distance: for a triangle (float)beta and (float)gamma, for a sphere the float ray
// parameter -- exactly the values the reference builds the hit point from (:329-330,
// :246-255), so the hit record needs no second intersection test.
struct HitUV {
    float a, b;
};

// ---------------------------------------------------------------- hit record
// Re-evaluates the winning primitive with the same arithmetic and writes the
// Intersect fields the shading needs (src/raytrace.cpp:246-255, :329-387).
__device__ __forceinline__ void make_hit(const DScene& S, F3 o, F3 d, double dd, int prim, float dmax,
                                F3& point, F3& normal, int& mat) {
    if (prim < S.ns) {
        float4 s = S.sph[prim];
        float d2, tf;
        sphere_test(o, d, dd, s, dmax, d2, tf);
        point = o + (d * tf);
        normal = norm3(point - f3(s));
        mat = S.sph_mat[prim];
    } else {
        int t = prim - S.ns;
        F3 a, ab, ac;
        tri_load(S.tri + 3 * (size_t)t, a, ab, ac);
        float4 q2 = S.tri[3 * (size_t)t + 2];
        TriHit h;
        h.beta = h.gamma = 0; h.point = f3(0, 0, 0); h.d2 = 0;
        tri_test(o, d, a, ab, ac, dmax, h);
        point = h.point;
        normal = tri_normal(S.tri_n + 3 * (size_t)t, (int)fbits(q2.z), d, h);
        mat = (int)fbits(q2.y);
    }
}

// Same record from the values the walk kept.  Smooth triangles blend their normals with
// the DOUBLE barycentrics (src/raytrace.cpp:344-357), so they are re-tested.
__device__ __forceinline__ void make_hit_uv(const DScene& S, F3 o, F3 d, int prim, HitUV uv, F3& point, F3& normal,
                                   int& mat) {
    if (prim < S.ns) {
        float4 s = S.sph[prim];
        point = o + (d * uv.a);
        normal = norm3(point - f3(s));
        mat = S.sph_mat[prim];
    } else {
        int t = prim - S.ns;
        F3 a, ab, ac;
        tri_load(S.tri + 3 * (size_t)t, a, ab, ac);
        float4 q2 = S.tri[3 * (size_t)t + 2];
        mat = (int)fbits(q2.y);
        if (fbits(q2.z) == 0u) {
            point = a + ((neg3(ab) * uv.a) + (neg3(ac) * uv.b));
            F3 n0 = f3(S.tri_n[3 * (size_t)t]);
            double ang = (double)dot3(n0, d);
            normal = (ang > 0) ? neg3(n0) : n0;
        } else {
            TriHit h;
            h.beta = h.gamma = 0; h.point = f3(0, 0, 0); h.d2 = 0;
            tri_test(o, d, a, ab, ac, __builtin_inff(), h);
            point = h.point;
            normal = tri_normal(S.tri_n + 3 * (size_t)t, 1, d, h);
        }
    }
}

// ---------------------------------------------------------------- lights
// Direction to the light and squared distance bound of the shadow query
// (src/raytrace.cpp:523-534).
__device__ __forceinline__ void light_dir(const DLight& L, F3 point, F3& I, float& dist2) {
    if (L.type == 1) {                       // DIRECTIONAL
        I = norm3(f3(L.dx, L.dy, L.dz) * -1.0f);
        dist2 = __builtin_inff();
    } else {                                 // POINT, SPOT
        F3 v = f3(L.px, L.py, L.pz) - point;
        dist2 = magsq3(v);
        I = norm3(v);
    }
}
