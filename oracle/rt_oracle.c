/* TEST INFRASTRUCTURE ONLY -- see rt_oracle.h.
 *
 * Every function cites the reference lines it restates.  Arithmetic rules that
 * hold throughout (from src/parse.h:25-75, src/parse.cpp:301-396):
 *   - a vector is three floats; vector (+,-) vector and vector (*,/) scalar are
 *     float operations, and a double scalar is narrowed to float at the call;
 *   - dot / cross / magnitudeSq / lengthSq are float, summed left to right;
 *   - magnitude() = (float)sqrt((double)sumsq); norm() = three float divisions.
 * Build with -ffp-contract=off so no product/sum is fused.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vld(const float* p) { return V(p[0], p[1], p[2]); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, float c) { return V(a.x * c, a.y * c, a.z * c); }
static inline v3 vdiv(v3 a, float c) { return V(a.x / c, a.y / c, a.z / c); }
static inline float vdot(v3 u, v3 v) { return (u.x * v.x) + (u.y * v.y) + (u.z * v.z); }
static inline v3 vcross(v3 u, v3 v) {
    return V(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
static inline float vmagsq(v3 a) { return (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
static inline float vlensq(v3 u, v3 v) { return vmagsq(vsub(u, v)); }
static inline float vmag(v3 a) { return (float)sqrt((double)vmagsq(a)); }
static inline v3 vnorm(v3 a) { float m = vmag(a); return V(a.x / m, a.y / m, a.z / m); }

/* material record: amb3 dif3 spec3 ns trans3 ior (scene-file order, src/parse.cpp:139-149) */
typedef struct { v3 amb, dif, spec, trans; float ns, ior; } mat_t;
static mat_t mat_at(const orc_scene* s, int i) {
    const float* m = s->materials + 14 * (size_t)i;
    mat_t r;
    r.amb = vld(m); r.dif = vld(m + 3); r.spec = vld(m + 6); r.ns = m[9];
    r.trans = vld(m + 10); r.ior = m[13];
    return r;
}

typedef struct {
    v3 normal, rp, rd, p;
    double dist;
    int mat;
    int kind; /* 1 sphere, 2 triangle */
    int prim;
} hit_t;

/* ------------------------------------------------------------------ jitter */
static inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
uint32_t orc_jitter_hash(uint32_t seed, uint64_t counter) {
    uint32_t hi = lowbias32((uint32_t)(counter >> 32) + seed * 0x9E3779B9U + 0x85EBCA6BU);
    return lowbias32((uint32_t)counter ^ hi) >> 1;
}

/* ------------------------------------------------------------------ camera */
/* Scene::getPlaneDist (src/parse.cpp:275-277): double quotient returned as float. */
static float plane_dist(const orc_scene* s) {
    return (float)(s->height / (2.0 * tan((double)s->cam_half_angle)));
}

/* RayTrace::getExtremePoints (src/raytrace.cpp:635-647) */
static void extreme_points(const orc_scene* s, double d, double w, double h, v3* p1, v3* p2) {
    v3 p0 = vld(s->cam_pos);
    v3 up = vld(s->cam_up);
    v3 dir = vld(s->cam_dir);
    v3 right = vcross(dir, up);
    p0 = vadd(p0, vmul(dir, (float)d));
    *p1 = vadd(p0, vmul(right, (float)(w / 2.0)));
    *p2 = vsub(p0, vmul(right, (float)(w / 2.0)));
    *p1 = vadd(*p1, vmul(up, (float)(h / 2.0)));
    *p2 = vsub(*p2, vmul(up, (float)(h / 2.0)));
}

void orc_camera_frame(const orc_scene* s, double out[7]) {
    v3 p1, p2;
    double pd = plane_dist(s);
    extreme_points(s, pd, s->width, s->height, &p1, &p2);
    out[0] = pd;
    out[1] = p1.x; out[2] = p1.y; out[3] = p1.z;
    out[4] = p2.x; out[5] = p2.y; out[6] = p2.z;
}

/* RayTrace::getRay (src/raytrace.cpp:441-485); r1,r2 from the shared jitter
 * sequence instead of rand() when sampling. */
static void get_ray(const orc_scene* s, int x, int y, int j, v3 p1, v3 p2, v3* ro, v3* rd) {
    int w = s->width, h = s->height;
    v3 pos = vld(s->cam_pos);
    v3 cdir = vld(s->cam_dir);
    v3 cup = vld(s->cam_up);
    v3 right = vnorm(vcross(cdir, cup));
    v3 down = vnorm(vmul(cup, -1.0f));
    double r1, r2;
    if (s->spp > 1) {
        uint64_t c = (((uint64_t)y * (uint64_t)w + (uint64_t)x) * (uint64_t)s->spp + (uint64_t)j) * 2u;
        r1 = (double)(int)orc_jitter_hash(s->seed, c) / 2147483647;
        r2 = (double)(int)orc_jitter_hash(s->seed, c + 1) / 2147483647;
    } else {
        r1 = r2 = .5;
    }
    v3 ptop = vsub(p2, p1);
    v3 ur = vmul(right, vdot(ptop, right));
    ur = vmul(ur, (float)((double)x / w + (r1 / w)));
    v3 bl = vmul(down, vdot(ptop, down));
    bl = vmul(bl, (float)((double)y / h + (r2 / h)));
    v3 p = vadd(ur, bl);
    p = vadd(p, p1);
    if (s->proj == ORC_PERSP) {
        *ro = pos;
        *rd = vnorm(p);
    } else {
        *ro = p;
        *rd = cdir;
    }
}

void orc_get_ray(const orc_scene* s, int x, int y, int j, float o[3], float d[3]) {
    v3 p1, p2, ro, rd;
    double pd = plane_dist(s);
    extreme_points(s, pd, s->width, s->height, &p1, &p2);
    get_ray(s, x, y, j, p1, p2, &ro, &rd);
    o[0] = ro.x; o[1] = ro.y; o[2] = ro.z;
    d[0] = rd.x; d[1] = rd.y; d[2] = rd.z;
}

/* ------------------------------------------------------------- intersection */
/* RayTrace::intersectSpheres (src/raytrace.cpp:216-302) */
static int intersect_spheres(const orc_scene* s, v3 posit, v3 direc, double dmin, double dmax,
                             hit_t* in) {
    int found = 0;
    int n = s->n_spheres;
    if (n == 0) return 0;
    double dd = vdot(direc, direc);
    for (int i = 0; i < n; i++) {
        const float* sp = s->spheres + 4 * (size_t)i;
        v3 c = vld(sp);
        float rad = sp[3];
        v3 ec = vsub(posit, c);
        double dec = vdot(direc, ec);
        double det = (dec * dec) - dd * (vdot(ec, ec) - (rad * rad));
        if (det < 0.0) continue;
        double sq = sqrt(det);
        double t1 = (-dec + sq) / dd;
        double t2 = (-dec - sq) / dd;
        v3 q1 = vmul(direc, (float)t1);
        v3 q2 = vmul(direc, (float)t2);
        double d1 = vmagsq(q1);
        double d2 = vmagsq(q2);
        /* which root is tried first depends on t1<t2 (:244, :272); the first
         * that passes its window test becomes this sphere's candidate */
        double ta, tb, da, db; v3 qa, qb;
        if (t1 < t2) { ta = t1; da = d1; qa = q1; tb = t2; db = d2; qb = q2; }
        else         { ta = t2; da = d2; qa = q2; tb = t1; db = d1; qb = q1; }
        double dc; v3 qc;
        if (ta > 0 && da > dmin && da < dmax) { dc = da; qc = qa; }
        else if (tb > 0 && db > dmin && db < dmax) { dc = db; qc = qb; }
        else continue;
        if (!found || dc < in->dist) {
            v3 point = vadd(posit, qc);
            in->normal = vnorm(vsub(point, c));
            in->rp = posit; in->rd = direc;
            in->dist = dc;
            in->p = point;
            in->mat = s->sphere_mat[i];
            in->kind = 1; in->prim = i;
            found = 1;
        }
    }
    return found;
}

/* RayTrace::intersectTriangle (src/raytrace.cpp:304-393) */
static int intersect_triangles(const orc_scene* s, v3 rp, v3 rd, double dmin, double dmax,
                               hit_t* in) {
    int found = 0;
    int n = s->n_triangles;
    for (int i = 0; i < n; i++) {
        const float* tv = s->tri_verts + 9 * (size_t)i;
        v3 v0 = vld(tv), v1 = vld(tv + 3), v2 = vld(tv + 6);
        v3 ae = vsub(v0, rp);
        v3 ab = vsub(v0, v1);
        v3 ac = vsub(v0, v2);
        double detA = ab.x * (ac.y * rd.z - rd.y * ac.z) + ab.y * (ac.z * rd.x - rd.z * ac.x) +
                      ab.z * (ac.x * rd.y - ac.y * rd.x);
        double dett = ac.z * (ab.x * ae.y - ab.y * ae.x) + ac.y * (ae.x * ab.z - ab.x * ae.z) +
                      ac.x * (ab.y * ae.z - ab.z * ae.y);
        double tval = dett / detA;
        if (tval > 0) continue;
        double detb = ae.x * (ac.y * rd.z - rd.y * ac.z) + ae.y * (ac.z * rd.x - rd.z * ac.x) +
                      ae.z * (ac.x * rd.y - ac.y * rd.x);
        double detg = rd.z * (ab.x * ae.y - ae.x * ab.y) + rd.y * (ab.z * ae.x - ae.z * ab.x) +
                      rd.x * (ab.y * ae.z - ae.y * ab.z);
        double beta = detb / detA;
        double gamma = detg / detA;
        if (beta >= 0 && beta <= 1 && gamma >= 0 && gamma <= 1 && beta + gamma <= 1) {
            v3 point = vadd(v0, vadd(vmul(vsub(v1, v0), (float)beta), vmul(vsub(v2, v0), (float)gamma)));
            double dist = vlensq(point, rp);
            if (dist < dmin || dist > dmax) continue;
            if (!found || dist < in->dist) {
                const float* tn = s->tri_normals + 9 * (size_t)i;
                v3 n0 = vld(tn), n1 = vld(tn + 3), n2 = vld(tn + 6);
                v3 normal;
                if (s->tri_flags[i]) {
                    double alpha = 1.0 - (beta + gamma);
                    double ang1 = vdot(n0, rd);
                    double ang2 = vdot(n1, rd);
                    double ang3 = vdot(n2, rd);
                    v3 m1 = (ang1 > 0) ? vmul(n0, -1.0f) : n0;
                    v3 m2 = (ang2 > 0) ? vmul(n1, -1.0f) : n1;
                    v3 m3 = (ang3 > 0) ? vmul(n2, -1.0f) : n2;
                    normal = vnorm(vadd(vadd(vmul(m1, (float)beta), vmul(m2, (float)gamma)),
                                        vmul(m3, (float)alpha)));
                } else {
                    double ang = vdot(n0, rd);
                    normal = (ang > 0) ? vmul(n0, -1.0f) : n0;
                }
                in->normal = normal;
                in->rp = rp; in->rd = rd;
                in->dist = dist;
                in->p = point;
                in->mat = s->tri_mat[i];
                in->kind = 2; in->prim = i;
                found = 1;
            }
        }
    }
    return found;
}

/* RayTrace::intersect, !useBVH branch (src/raytrace.cpp:154-181) */
static int intersect(const orc_scene* s, v3 rp, v3 rd, double dmin, double dmax, hit_t* out) {
    hit_t sph, tri;
    int hs = intersect_spheres(s, rp, rd, dmin, dmax, &sph);
    int ht = intersect_triangles(s, rp, rd, dmin, dmax, &tri);
    int have = 0;
    if (hs) { *out = sph; have = 1; }
    if (ht) {
        if (!have) { *out = tri; have = 1; }
        else if (out->dist > tri.dist) { *out = tri; }
    }
    return have;
}

int orc_intersect(const orc_scene* s, const float o[3], const float d[3], double dmin, double dmax,
                  double out[8]) {
    hit_t h;
    if (!intersect(s, vld(o), vld(d), dmin, dmax, &h)) return 0;
    out[0] = h.normal.x; out[1] = h.normal.y; out[2] = h.normal.z;
    out[3] = h.dist;
    out[4] = h.p.x; out[5] = h.p.y; out[6] = h.p.z;
    out[7] = h.prim;
    return h.kind;
}

/* ------------------------------------------------------------------ shading */
static v3 eval_ray_tree(const orc_scene* s, v3 rp, v3 rd, int depth, orc_counts* cnt);

/* How often each branch of the spot-light fall-off (src/raytrace.cpp:552-565) was taken since the
 * last reset: [0] alpha < angle1, [1] alpha > angle2, [2] interpolated.  Lets a test prove that a
 * scene exercises all three. */
static unsigned long long g_spot_branch[3];
void orc_spot_branch_counts(unsigned long long out[3], int reset) {
    for (int k = 0; k < 3; k++) {
        out[k] = g_spot_branch[k];
        if (reset) g_spot_branch[k] = 0;
    }
}
static void spot_branch(int k) {
#ifdef _OPENMP
    #pragma omp atomic
#endif
    g_spot_branch[k]++;
}

/* RayTrace::getColor (src/raytrace.cpp:488-632) */
static v3 get_color(const orc_scene* s, const hit_t* i, int depth, orc_counts* cnt) {
    mat_t m = mat_at(s, i->mat);
    v3 dif = m.dif, spec = m.spec;
    float rval, gval, bval;
    rval = m.amb.x; gval = m.amb.y; bval = m.amb.z;
    v3 point = i->p;
    v3 direct = i->rd;
    v3 normal = i->normal;
    rval *= s->ambient[0];
    gval *= s->ambient[1];
    bval *= s->ambient[2];
    for (int j = 0; j < s->n_lights; j++) {
        const orc_light* L = &s->lights[j];
        v3 I;
        double dist;
        if (L->type == ORC_DIRECTIONAL) {
            I = vmul(vld(L->dir), (float)-1.0);
            I = vnorm(I);
            dist = INFINITY;
        } else {
            I = vsub(vld(L->pos), point);
            dist = vmagsq(I);
            I = vnorm(I);
        }
        hit_t sh;
        if (cnt) cnt->shadow++;
        if (intersect(s, point, I, .001, dist, &sh)) continue;
        double dotni = vdot(normal, I);
        double cosAlpha = (dotni > 0) ? dotni : 0;
        v3 illum = vld(L->color);
        if (L->type == ORC_POINT) {
            illum = vmul(illum, (float)(1.0 / dist));
        } else if (L->type == ORC_SPOT) {
            double alpha = acos(cosAlpha);
            if (alpha < L->angle1) {
                spot_branch(0);
                illum = vmul(vld(L->color), (float)(1.0 / dist));
            } else if (alpha > L->angle2) {
                spot_branch(1);
                illum = V(0, 0, 0);
            } else {
                spot_branch(2);
                double amt = 1 - ((alpha - L->angle1) / (L->angle2 - L->angle1));
                illum = vmul(vld(L->color), (float)((1.0 / dist) * amt));
            }
        }
        rval += dif.x * illum.x * cosAlpha;
        gval += dif.y * illum.y * cosAlpha;
        bval += dif.z * illum.z * cosAlpha;
        v3 ref = vsub(vmul(normal, (float)(2.0 * dotni)), I);
        v3 Vv = vmul(direct, -1.0f);
        double powval = vdot(vnorm(Vv), vnorm(ref));
        double pspec;
        if (powval < 0.0) pspec = 0.0;
        else pspec = pow(powval, (double)m.ns);
        rval += spec.x * pspec * illum.x;
        gval += spec.y * pspec * illum.y;
        bval += spec.z * pspec * illum.z;
    }
    v3 rdir = direct;
    v3 irdir = vmul(rdir, -1.0f);
    if (spec.x != 0 && spec.y != 0 && spec.z != 0) {
        v3 rfd = vsub(vmul(normal, 2.0f * vdot(normal, irdir)), irdir);
        v3 rc = eval_ray_tree(s, point, rfd, depth + 1, cnt);
        rval += spec.x * rc.x;
        gval += spec.y * rc.y;
        bval += spec.z * rc.z;
    }
    v3 trans = m.trans;
    if (trans.x != 0 && trans.y != 0 && trans.z != 0) {
        double ior;
        double dni = vdot(irdir, normal);
        if (dni <= 0) ior = m.ior;
        else ior = 1.0 / m.ior;
        double tir = 1.0 - (ior * ior) * (1.0 - (dni * dni));
        if (tir >= 0) {
            v3 refdir;
            if (dni >= 0) refdir = vsub(vmul(normal, (float)(ior * dni - sqrt(tir))), vmul(irdir, (float)ior));
            else          refdir = vsub(vmul(normal, (float)(ior * dni + sqrt(tir))), vmul(irdir, (float)ior));
            v3 rc = eval_ray_tree(s, point, vnorm(refdir), depth + 1, cnt);
            rval += trans.x * rc.x;
            gval += trans.y * rc.y;
            bval += trans.z * rc.z;
        }
    }
    return V(rval, gval, bval);
}

/* RayTrace::evaluateRayTree (src/raytrace.cpp:131-151) */
static v3 eval_ray_tree(const orc_scene* s, v3 rp, v3 rd, int depth, orc_counts* cnt) {
    if (depth > s->max_depth) return vld(s->background);
    hit_t in;
    if (cnt) { if (depth == 0) cnt->primary++; else cnt->secondary++; }
    if (intersect(s, rp, rd, .001, INFINITY, &in)) return get_color(s, &in, depth, cnt);
    return vld(s->background);
}

void orc_trace_ray(const orc_scene* s, const float o[3], const float d[3], int depth, float rgb[3],
                   orc_counts* counts) {
    v3 c = eval_ray_tree(s, vld(o), vld(d), depth, counts);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ---------------------------------------------------------------- quantise */
/* (int)(double) as x86-64 cvttsd2si performs it: NaN and out-of-range give
 * INT_MIN; then ComponentClamp (src/pixel.h:17-18). */
static uint8_t quant1(float c) {
    double v = c * 255.0;
    int i;
    if (!(v == v) || v >= 2147483648.0 || v <= -2147483649.0) i = (-2147483647 - 1);
    else i = (int)v;
    return (uint8_t)((i < 0) ? 0 : (i > 255) ? 255 : i);
}
void orc_quantise(const float rgb[3], uint8_t out[3]) {
    out[0] = quant1(rgb[0]); out[1] = quant1(rgb[1]); out[2] = quant1(rgb[2]);
}

/* RayTrace::ave (src/raytrace.cpp:1079-1089) */
static v3 ave(const v3* v, int num) {
    if (num == 1) return v[0];
    v3 u = v[0];
    for (int i = 1; i < num; i++) u = vadd(u, v[i]);
    return vdiv(u, (float)num);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* pixel loop of RayTrace::rayTrace (src/raytrace.cpp:83-122), x outer / y inner,
 * OpenMP collapse(2) schedule(static,125) as the reference's PARALLEL build; a region too small to give every
 * thread such a chunk (the crops the tests compare) is handed out pixel by pixel instead -- pixels do not depend on
 * one another, so the schedule cannot change a result. */
double orc_render(const orc_scene* s, int x0, int y0, int x1, int y1, int nthreads, uint8_t* rgb8,
                  float* rgbf, orc_counts* counts) {
    v3 p1, p2;
    double pd = plane_dist(s);
    extreme_points(s, pd, s->width, s->height, &p1, &p2);
    int spp = s->spp;
    int rw = x1 - x0;
    uint64_t cp = 0, cs = 0, csh = 0;
    double t0 = now_s();
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    if ((long long)(x1 - x0) * (y1 - y0) < 125LL * omp_get_max_threads()) omp_set_schedule(omp_sched_dynamic, 1);
    else omp_set_schedule(omp_sched_static, 125);
    #pragma omp parallel for collapse(2) schedule(runtime) reduction(+ : cp, cs, csh)
#endif
    for (int x = x0; x < x1; x++) {
        for (int y = y0; y < y1; y++) {
            orc_counts c = {0, 0, 0};
            v3 col;
            if (spp > 1) {
                v3* color = (v3*)malloc(sizeof(v3) * (size_t)spp);
                for (int j = 0; j < spp; j++) {
                    v3 ro, rd;
                    get_ray(s, x, y, j, p1, p2, &ro, &rd);
                    color[j] = eval_ray_tree(s, ro, rd, 0, &c);
                }
                col = ave(color, spp);
                free(color);
            } else {
                v3 ro, rd;
                get_ray(s, x, y, 0, p1, p2, &ro, &rd);
                col = eval_ray_tree(s, ro, rd, 0, &c);
            }
            size_t idx = (size_t)(y - y0) * rw + (x - x0);
            if (rgb8) { rgb8[3 * idx] = quant1(col.x); rgb8[3 * idx + 1] = quant1(col.y); rgb8[3 * idx + 2] = quant1(col.z); }
            if (rgbf) { rgbf[3 * idx] = col.x; rgbf[3 * idx + 1] = col.y; rgbf[3 * idx + 2] = col.z; }
            cp += c.primary; cs += c.secondary; csh += c.shadow;
        }
    }
    double t = now_s() - t0;
    if (counts) { counts->primary = cp; counts->secondary = cs; counts->shadow = csh; }
    return t;
}
