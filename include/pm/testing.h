/* Test hooks of libvehicle_pm_gpu.so -- NOT part of the supported C ABI (include/pm/patchmatch.h): entry points that
 * exist only so that tests/ can drive the library into states a well-behaved caller never produces.  They have no
 * counterpart in the reference (src/vehicle/patchmatch_gpu/patchmatch_gpu.h:77-124) and may change or vanish between
 * ABI versions; the C++ wrapper (host/patchmatch_gpu.hpp) does not use them. */
#ifndef PM_TESTING_H_
#define PM_TESTING_H_
#include "pm/patchmatch.h"
#ifdef __cplusplus
extern "C" {
#endif
/* forks an empty dependency onto an internal stream of an open capture and leaves it unjoined, so that the guard in
 * pm_capture_end (PM_ERR_STATE instead of a fault inside the runtime) can be exercised */
int pm_debug_capture_fork(pm_handle* h);

/* ---- the row-tiled driver's device discipline, provable on ONE GPU ------------------------------------------------
 * pm_tiled_create with the bands accounted to LOGICAL devices: band k lives on logical_devices[k] (>= 0; several bands
 * may share one) while every HIP call still goes to the physical device of the band's handle.  Such a plan logs every
 * runtime call pm_tiled.hip makes -- hipSetDevice, allocations, event creation / record / wait, copies, stream
 * synchronisation, and the pm_tile_* stages with their pointer arguments -- with the logical device that was current and
 * the logical devices its stream, event and pointers belong to, and marks what would be an error (or a silent cross-device
 * access) if the logical devices were physical ones.  simulate_peer_access: 1 = neighbouring bands on different logical
 * devices count as peer-linked (what hipDeviceEnablePeerAccess succeeding in both directions gives), so that
 * PM_TILED_EXCHANGE_DIRECT takes its cross-device path; 0 = they do not.  No counterpart in the reference (one GPU,
 * src/vehicle/patchmatch_gpu/patchmatch_gpu.cu:331-376). */
typedef enum pm_tiled_audit_call {
  PM_TILED_CALL_SET_DEVICE = 1,
  PM_TILED_CALL_MALLOC = 2,
  PM_TILED_CALL_EVENT_CREATE = 3,
  PM_TILED_CALL_EVENT_RECORD = 4,
  PM_TILED_CALL_STREAM_WAIT_EVENT = 5,
  PM_TILED_CALL_STREAM_SYNC = 6,
  PM_TILED_CALL_MEMSET = 7,
  PM_TILED_CALL_COPY_H2D = 8,
  PM_TILED_CALL_COPY_D2H = 9,
  PM_TILED_CALL_COPY_PEER = 10, /* hipMemcpyPeerAsync: the one runtime call that names both devices                   */
  PM_TILED_CALL_STAGE = 11,     /* a pm_tile_* stage: kernels on the band's stream (detail = stage id * 16)            */
  PM_TILED_CALL_STAGE_ARG = 12  /* one device pointer handed to that stage (detail = stage id * 16 + argument number)  */
} pm_tiled_audit_call;
enum { /* pm_tiled_audit_record.violation, a bit mask */
  PM_TILED_BAD_STREAM_DEVICE = 1,  /* a stream was used while another device was current                               */
  PM_TILED_BAD_EVENT_RECORD = 2,   /* an event was recorded on a stream of another device                              */
  PM_TILED_BAD_OBJECT_DEVICE = 4,  /* allocation / event creation / destination with another device current           */
  PM_TILED_BAD_FOREIGN_READ = 8,   /* a pointer into another device's memory where the call may not read there         */
  PM_TILED_BAD_UNKNOWN = 16        /* a stream, event or pointer the plan does not know                                */
};
typedef struct pm_tiled_audit_record {
  int call;            /* pm_tiled_audit_call                                                            */
  int band;            /* the band the call was made for                                                 */
  int detail;          /* COPY_PEER: the band that owns the source; STAGE / STAGE_ARG: see above          */
  int current_device;  /* logical device current at the call                                             */
  int stream_device;   /* logical device of the stream argument, -1: the call has none                   */
  int object_device;   /* ... of the event / the allocation / the destination pointer, -1: none          */
  int source_device;   /* ... of the source pointer (copies, stage arguments), -1: none                  */
  int foreign_allowed; /* 1: the call may read a source in another device's memory (peer copy, direct exchange) */
  int violation;       /* 0 = fine                                                                       */
} pm_tiled_audit_record;
int pm_tiled_create_logical(pm_handle* const* bands, int n_bands, int rows, int cols, const int* logical_devices,
                            int simulate_peer_access, pm_tiled_plan** out);
/* copies the first `capacity` records (records may be NULL), *total = records logged, *violations = marked ones */
int pm_tiled_audit(const pm_tiled_plan* plan, pm_tiled_audit_record* records, int capacity, int* total, int* violations);
int pm_tiled_audit_reset(pm_tiled_plan* plan);
/* Makes a logical-device plan BREAK the discipline on purpose at every boundary-row hand-over, so that a test can see the
 * log catch it: bit 0 = the reader records an event of the PUBLISHER's device on its own stream (what the driver did until
 * round 5); bit 1 = a stream is used while another band's device is current.  Harmless where all logical devices are one
 * physical device (the events involved are waited for by nobody); 0 switches it off. */
int pm_tiled_debug_inject(pm_tiled_plan* plan, int what);
#ifdef __cplusplus
}
#endif
#endif /* PM_TESTING_H_ */
