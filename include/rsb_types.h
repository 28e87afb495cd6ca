/*
 * rsb_types.h - the plain-data part of the C-ABI (include/rsb.h includes it): capacities, enums, the model blob, the contact
 * record, the state-field ids.  Split out because the HIP kernels compile against exactly this part: a change to rsb.h's function
 * declarations no longer rebuilds the ~90 kernel objects.
 */
#ifndef RSB_TYPES_H_
#define RSB_TYPES_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSB_MAX_BODIES 64      /* moving bodies incl. floating base                    */
#define RSB_MAX_DOF (6 + RSB_MAX_BODIES - 1)
#define RSB_MAX_COLLISIONS 64  /* collision spheres per articulated system             */
#define RSB_MAX_CONTACTS 16    /* upper bound on the per-env contact cap (k_max)       */
#define RSB_NAME_LEN 48

typedef enum rsb_status {
  RSB_OK = 0,
  RSB_E_INVALID = -1,    /* bad argument                                   */
  RSB_E_PARSE = -2,      /* URDF parse error                               */
  RSB_E_UNSUPPORTED = -3,/* feature outside the supported subset           */
  RSB_E_NO_DEVICE = -4,  /* no HIP device / HIP runtime failure            */
  RSB_E_HIP = -5,        /* a HIP call failed                              */
  RSB_E_STATE = -6,      /* call not valid in the current state            */
  RSB_E_PIPELINE = -7    /* pipelined control steps faulted on the device; the library has replayed them in lock-step (rsb_pipeline.h) */
} rsb_status;

typedef enum rsb_memspace { RSB_HOST = 0, RSB_DEVICE = 1 } rsb_memspace;

/* raisim::ControlMode::Type [RECALL] */
typedef enum rsb_control_mode {
  RSB_FORCE_AND_TORQUE = 0,
  RSB_PD_PLUS_FEEDFORWARD_TORQUE = 1
} rsb_control_mode;

typedef enum rsb_joint_type { RSB_JOINT_FLOATING = 0, RSB_JOINT_REVOLUTE = 1, RSB_JOINT_PRISMATIC = 2 } rsb_joint_type;

/*
 * Flat, immutable description of one articulated system (the "model blob").
 * Body 0 is the floating base; body i>0 is attached to parent[i] < i by a 1-DoF joint whose
 * frame sits at ptree[i] / rtree[i] in the parent body frame and moves about/along axis[i]
 * (expressed in the joint = child-body frame).  Fixed URDF joints are already merged.
 * Capsules are stored as their two end spheres (the contact set ODE's capsule-plane collider
 * produces); every collision primitive is therefore a sphere.
 */
typedef struct rsb_model_blob {
  int32_t nb, nq, nv, ncol, depth;
  int32_t fixed_base;   /* != 0: body 0 does not move (a URDF whose root link is named "world" [RECALL RaiSim's convention]); gc / gv keep their 7 / 6 base
                           entries (ignored on input, constant on output), the joints follow as usual */
  int32_t parent[RSB_MAX_BODIES];
  int32_t level[RSB_MAX_BODIES];
  int32_t jtype[RSB_MAX_BODIES];
  double axis[RSB_MAX_BODIES][3];
  double ptree[RSB_MAX_BODIES][3];
  double rtree[RSB_MAX_BODIES][9];   /* row-major, parent <- joint frame            */
  double mass[RSB_MAX_BODIES];
  double com[RSB_MAX_BODIES][3];     /* body frame                                  */
  double inertia[RSB_MAX_BODIES][6]; /* xx xy xz yy yz zz about com, body frame     */
  double armature[RSB_MAX_BODIES];   /* rotor inertia added to M's diagonal         */
  double damping[RSB_MAX_BODIES];    /* viscous joint damping                       */
  double q_lower[RSB_MAX_BODIES], q_upper[RSB_MAX_BODIES];
  double effort[RSB_MAX_BODIES];     /* |tau| limit, <=0 means unlimited            */
  int32_t col_body[RSB_MAX_COLLISIONS];
  double col_pos[RSB_MAX_COLLISIONS][3]; /* sphere centre, body frame                */
  double col_radius[RSB_MAX_COLLISIONS];
  char body_name[RSB_MAX_BODIES][RSB_NAME_LEN];   /* URDF link name of each moving body  */
  char joint_name[RSB_MAX_BODIES][RSB_NAME_LEN];  /* URDF joint name (index 0: "base")   */
  char col_name[RSB_MAX_COLLISIONS][RSB_NAME_LEN];
  /* rim primitives (the end caps of a <cylinder>): col_rim[s] > 0 makes primitive s the LOWEST POINT of the circle of that radius
   * around col_pos[s] in the plane normal to col_axis[s] (body frame) - lowest with respect to the terrain normal under the
   * centre; col_radius[s] is 0 for them.  col_rim[s] == 0: a sphere. */
  double col_axis[RSB_MAX_COLLISIONS][3];
  double col_rim[RSB_MAX_COLLISIONS];
  char col_material[RSB_MAX_COLLISIONS][RSB_NAME_LEN];  /* <collision><material name=".."/> of the URDF, "default" if absent */
  /* capsules and cylinders: col_capsule[s] = e + 1 makes primitives s and e the two ENDS of one capsule (two spheres of one radius) or of
   * one cylinder (two rim primitives of one rim radius) - same body; set on the first of the two, 0 everywhere else.  On a plane the two
   * ends are the exact contact set; against a height map the barrel between them can touch where neither end does (a shank lying
   * across a ridge): rsb_set_capsule_contacts.
   * boxes: col_capsule[s] = -1 makes primitives s .. s + 7 the eight CORNERS of one box (corner s + e: bit 0 of e = +x, bit 1 = +y, bit 2 = +z
   * in the box's frame; same body, radius 0).  On a plane the corners are the exact contact set; against a height map a face can touch
   * where no corner does (a slab lying on a bump): the same switch adds the deepest point of the box's faces. */
  int32_t col_capsule[RSB_MAX_COLLISIONS];
} rsb_model_blob;

/* One solved contact, as raisim::Contact exposes it (position/normal/impulse/body index). */
typedef struct rsb_contact {
  float position[3];   /* world frame                                          */
  float normal[3];     /* world frame, pointing from terrain into the robot    */
  float impulse[3];    /* world frame, impulse applied to the robot over dt    */
  float depth;
  int32_t body;        /* local body index of the articulated system           */
  int32_t collision;   /* collision primitive index; a self-collision is listed once per body (as raisim::Contact does:
                          isSelfCollision(), isObjectA()) and carries RSB_CONTACT_SELF_A / _B in this field: the two entries
                          sit next to each other, same position and depth, opposite normals and impulses */
} rsb_contact;
#define RSB_CONTACT_SECOND 0x40000   /* a primitive's second contact with a height map (rsb_set_heightmap_contacts) */
#define RSB_CONTACT_CAPSULE 0x80000  /* contact of the barrel of a capsule / cylinder (between its two ends) or of a face of a box (between its corners) with a height map; the id is the FIRST end's / corner's (rsb_set_capsule_contacts) */
#define RSB_CONTACT_SELF_A 0x10000
#define RSB_CONTACT_SELF_B 0x20000
#define RSB_CONTACT_PRIMITIVE(c) ((c) & 0xffff)

/* resident state fields (row-major [N,dim] float32 unless noted) */
typedef enum rsb_field {
  RSB_F_GC = 0, RSB_F_GV = 1, RSB_F_PTARGET = 2, RSB_F_DTARGET = 3, RSB_F_TAU_FF = 4,
  RSB_F_CONTACT_COUNT = 5, RSB_F_CONTACTS = 6, RSB_F_FLAGS = 7,
  RSB_F_GENERALIZED_FORCE = 8   /* output only, see rsb_enable_generalized_force_output */
} rsb_field;

#define RSB_MAX_RANKS 8           /* ranks of one node (peer-mapped obs exchange, rsb.h) */

#ifdef __cplusplus
}
#endif
#endif
