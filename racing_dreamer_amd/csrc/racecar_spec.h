// Env spec constants shared by host and device code.
// The CPU oracle (oracle/) restates the same numbers independently.
#pragma once

#define RCS_DT 0.01f              // dreamer/callbacks.py:23
#define RCS_INV_DT 100.0f
#define RCS_MAX_RANGE 15.0f       // dreamer/tools.py:274
#define RCS_LIDAR_X 0.25f
#define RCS_WHEELBASE 0.3302f     // ros_agent/agents/follow_the_gap/src/agent.py:78
#define RCS_MAX_STEER 0.42f       // ros_agent/models/dreamer/racing_dreamer.py:14 (the nominal scale of the steering action)
#define RCS_WHEEL_MAX 0.19f       // front-wheel angle at full command - and +1 steers RIGHT: both pinned by the reference's trained
#define RCS_STEER_GAIN -0.19f     // agents (tests/test_golden_policy.py): command -> wheel angle, counter-clockwise positive
#define RCS_MAX_VEL 5.0f          // ros_agent/models/dreamer/racing_dreamer.py:16
#define RCS_ACCEL_MAX 4.0f        // max_force 0.5 (racing_dreamer.py:15) * 8 m/s^2 per unit force
#define RCS_DRAG 0.8f             // 1/s = ACCEL_MAX / MAX_VEL: full throttle settles at max_velocity
#define RCS_STEER_STEP 0.032f     // 3.2 rad/s * dt
#define RCS_BOX_CX 0.175f         // car rectangle centre ahead of the rear axle
#define RCS_BOX_HL 0.275f         // half length
#define RCS_BOX_HW 0.15f          // half width
#define RCS_N_CHECKPOINTS 20
#define RCS_PROGRESS_REWARD 100.0f
#define RCS_PATCH_CELLS 3.125f    // 200 cells / 64 px            (dreamer/wrappers.py:402-405)
#define RCS_PATCH_WINDOW 110.0f   // neigh_size + 10 cells        (dreamer/wrappers.py:398-399)
#define RCS_PATCH_WINDOW_I 110
#define RCS_PATCH_STEP_Q16 204800.0f   // 3.125 cells per pixel in 16.16 fixed point
#define RCS_BALL_GAP_BINS 12      // 1.2 m between the cars of one env at reset
#define RCS_GRID_LEAD_BINS 8      // grid mode: the last car starts 0.8 m after the start line
#define RCS_SPAWN_CLEAR_R 40      // random starts: cells searched around a centre-line point for the nearest non-drivable cell
#define RCS_SPAWN_MARGIN 0.60f    // [m] the footprint's farthest corner (0.474) + the two half cell diagonals (0.071)
#define RCS_SPAWN_W_MAX 1.5f      // [m] cap of the lateral offset
#define RCS_HEADING_JITTER 0.35f  // [rad] heading within +- this of the track's direction - where the track leaves lateral room;
#define RCS_SPAWN_FOOT_R 5        // where it leaves none: by the footprint's own clearance k = 0 .. 5 cells (oracle: HEADING_ROOM)
#define RCS_HEADING_ROOM_INIT {0.0f, 0.0f, 0.05f, 0.155f, 0.26f, 0.35f}
#define RCS_SPAWN_SAFE_SEARCH 256 // several cars: bins searched forward for a start whose four centre-line poses do not overlap
#define RCS_N_FOOTPRINT 34
#define RCS_FOOT_STEP 0.05f      // pitch of the footprint lattice [m] (12 x 7 nodes, rear axle at node (2, 3))
#define RCS_PI 3.14159274101257324f
#define RCS_TWO_PI 6.28318548202514648f
