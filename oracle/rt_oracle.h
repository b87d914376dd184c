/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's per-pixel trace loop (LastingLuck/Raytracer,
 * src/raytrace.cpp, brute-force branch), written from the behaviour of that
 * file, operation for operation (float vs double placement, evaluation order,
 * no fused multiply-add).  It exists to CHECK the HIP path; nothing in the
 * product may include, link or call it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Pinning: tests/test_oracle_vs_reference.py compares this file bit-for-bit with
 * the unmodified reference (oracle/_ref/libref_oracle.so, built from
 * /root/reference/src by oracle/Makefile) on every shipped scene, and
 * tests/golden/ holds vectors generated from that reference build, so the
 * restatement stays pinned where the reference tree is absent.
 */
#ifndef RT_ORACLE_H_
#define RT_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* light types: values of the reference's enum LightType (src/parse.h:18-23) */
enum { ORC_AMBIENT = 0, ORC_DIRECTIONAL = 1, ORC_POINT = 2, ORC_SPOT = 3 };
/* projection: values of the reference's enum ProjType (src/parse.h:13-16) */
enum { ORC_ORTHO = 0, ORC_PERSP = 1 };

typedef struct orc_light {
    int32_t type;
    float color[3];
    float pos[3];
    float dir[3];
    float angle1, angle2;
} orc_light;

/* Same field order and meaning as rt_scene_desc in include/rt_api.h, declared
 * separately so the checker shares no code with the thing it checks. */
typedef struct orc_scene {
    int32_t width, height, spp, max_depth, proj;
    uint32_t seed;
    float cam_pos[3], cam_dir[3], cam_up[3];
    float cam_half_angle;
    float background[3];
    float ambient[3];
    int32_t n_spheres;
    int32_t n_triangles;
    int32_t n_materials;
    int32_t n_lights;
    const float* spheres;      /* [n_spheres][4]  cx cy cz r            */
    const int32_t* sphere_mat; /* [n_spheres]                            */
    const float* tri_verts;    /* [n_triangles][9]                       */
    const float* tri_normals;  /* [n_triangles][9]                       */
    const int32_t* tri_flags;  /* [n_triangles]  1 = normal_triangle     */
    const int32_t* tri_mat;    /* [n_triangles]                          */
    const float* materials;    /* [n_materials][14] amb3 dif3 spec3 ns trans3 ior */
    const orc_light* lights;   /* [n_lights]                             */
} orc_scene;

typedef struct orc_counts {
    uint64_t primary, secondary, shadow;
} orc_counts;

/* shared seeded jitter: value that stands in for rand() (0 .. 2^31-1) */
uint32_t orc_jitter_hash(uint32_t seed, uint64_t counter);

/* planeDist, p1[3], p2[3]  (src/parse.cpp:275-277, src/raytrace.cpp:635-647) */
void orc_camera_frame(const orc_scene* s, double out[7]);

/* primary ray of pixel (x,y) sample j  (src/raytrace.cpp:441-485) */
void orc_get_ray(const orc_scene* s, int x, int y, int j, float o[3], float d[3]);

/* RayTrace::intersect brute-force branch (src/raytrace.cpp:154-181).
 * returns 0 miss, 1 sphere hit, 2 triangle hit; out = normal3, dist2, point3, prim index */
int orc_intersect(const orc_scene* s, const float o[3], const float d[3], double dmin, double dmax,
                  double out[8]);

/* evaluateRayTree (src/raytrace.cpp:131-151) */
void orc_trace_ray(const orc_scene* s, const float o[3], const float d[3], int depth, float rgb[3],
                   orc_counts* counts);

/* times each branch of the spot-light fall-off ran since the last reset
 * (src/raytrace.cpp:552 alpha < angle1, :556 alpha > angle2, :560 interpolated) */
void orc_spot_branch_counts(unsigned long long out[3], int reset);

/* Pixel::SetClamp(c*255.0) (src/pixel.cpp:38-43, src/pixel.h:17-18) */
void orc_quantise(const float rgb[3], uint8_t out[3]);

/* pixel loop of RayTrace::rayTrace (src/raytrace.cpp:95-122) over ROI [x0,x1)x[y0,y1);
 * rgb8 / rgbf are ROI-sized, top row first; either may be NULL.  Returns wall seconds. */
double orc_render(const orc_scene* s, int x0, int y0, int x1, int y1, int nthreads, uint8_t* rgb8,
                  float* rgbf, orc_counts* counts);

#ifdef __cplusplus
}
#endif
#endif
