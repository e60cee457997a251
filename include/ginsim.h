/*
 * ginsim.h -- C ABI of the MI355X-native Monte-Carlo strapdown-INS engine (libginsim.so).
 *
 * This is the drop-in boundary for ONE hot path of Aceinna/gnss-ins-sim:
 *     pathgen.path_gen  ->  acc_gen / gyro_gen / bias_drift  ->  FreeIntegration.run  ->  end-point stats
 * Style mirrors the reference's own FFI precedent (demo_algorithms/mag_calibrate.py:44,78-86 and
 * demo_algorithms/aceinna_ins.py:172-237): extern "C" functions, POD structs by pointer, caller-owned
 * row-major double buffers, explicit sizes.  Unlike the precedent every function returns an int status
 * (0 = ok, <0 = error) and never aborts; ginsim_last_error() returns the thread-local message.
 *
 * All reference citations are relative to the reference repository root.
 */
#ifndef GINSIM_H
#define GINSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GINSIM_ABI_VERSION 8

/* status codes */
#define GINSIM_OK          0
#define GINSIM_ERR_ARG    -1   /* bad argument (Python wrapper raises ValueError, like the reference) */
#define GINSIM_ERR_HIP    -2   /* HIP runtime error (message holds hipGetErrorString) */
#define GINSIM_ERR_NODEV  -3   /* no usable GPU */
#define GINSIM_ERR_RANGE  -4   /* output capacity too small */
#define GINSIM_ERR_NOMEM  -5   /* ABI 7: the device is out of memory (hipErrorOutOfMemory): the one failure a caller may retry after
                                * giving memory back; every other HIP failure stays GINSIM_ERR_HIP */
#define GINSIM_ERR_PLACED -6   /* ABI 7: ginsim_malloc_placed / _reserve: this device has no usable placed arena (no virtual-memory
                                * management, fewer than two classes of physical memory found, arena limit reached): allocate with
                                * ginsim_malloc instead */

typedef struct ginsim_ctx ginsim_ctx;   /* one context per device; re-entrant per context */

/* ---- context / plumbing -------------------------------------------------------------------- */
int         ginsim_abi_version(void);
const char* ginsim_last_error(void);
int  ginsim_device_count(int* count);
int  ginsim_create(int device, ginsim_ctx** out);
/* ABI 8: the XCD (accelerator complex die, 0 .. 7 on an MI355X) on which the FIRST workgroup of a launch on this context's stream
 * lands.  The dispatcher deals the workgroups of a launch to the XCDs in turn, starting at a die that belongs to the stream's
 * hardware queue (measured: the same for every launch of a stream, consecutive for streams made one after the other).  Two launches
 * that are to run side by side without costing each other a round of workgroups must fit TOGETHER into whole rounds on every die:
 * the drop-in Sim runs the few runs it keeps as ONE workgroup next to the statistics launch over all the others (1023 workgroups on
 * 8 x 64 slots: one die has a slot to spare) and picks the pair of contexts that puts that workgroup on that die. */
int  ginsim_stream_first_xcc(ginsim_ctx* ctx, int32_t* xcc);
int  ginsim_destroy(ginsim_ctx* ctx);
int  ginsim_device_name(ginsim_ctx* ctx, char* buf, size_t cap);
int  ginsim_mem_info(ginsim_ctx* ctx, size_t* free_bytes, size_t* total_bytes);   /* ABI 6: hipMemGetInfo of the context's device */
int  ginsim_malloc(ginsim_ctx* ctx, size_t bytes, void** dptr);
int  ginsim_free(ginsim_ctx* ctx, void* dptr);   /* regions of ginsim_malloc and of ginsim_malloc_placed alike */

/* ---- ABI 7: placed device memory -------------------------------------------------------------------------------------------
 * The 288 GB of an MI355X are three 96 GB classes of physical memory (the top level of the physical address; below it every HBM
 * stack and channel is interleaved).  A launch that streams SEVERAL output planes at once -- the 15 planes [component][sample][run]
 * the fused kernel writes where the reference keeps R x (n,3) arrays per series (ins_sim.py:490-506, ins_algo_manager.py:77-95) --
 * writes 5.7-5.9 TB/s when all of them lie in one class, 6.4-6.5 TB/s across two and 6.8-7.0 TB/s across three
 * (profiles/r06_placed_memory.json), and hipMalloc gives a process its first tens of GB from one class.  hipMalloc does not say
 * where a region lies, but the virtual-memory API lets the library build a range from physical chunks it has looked at: a
 * placed arena (one per device, shared by the contexts of a process) is a reserved virtual range whose 512 MiB stripes are
 * hipMemCreate'd chunks dealt to the range so that consecutive stripes cycle through the classes.  A chunk's class is found by
 * TIMING a fill that streams into it and into a reference chunk of each class at once (two streams in one class conflict: the pair
 * takes 1.3-1.9 x the time of a pair in different classes).  Chunks are created until every class has its share of the request
 * (or the budget is spent: then two classes, or GINSIM_ERR_PLACED), the rest is given back to the driver.  Every region carved
 * from the arena larger than a stripe spans the classes, whatever its planes' sizes; regions are handed out by a first-fit free
 * list and the arena grows by whole stripes.  ginsim_destroy returns what its context carved and never freed; the device's last
 * context gives the arena's memory back.  The reference has no counterpart (its arrays are NumPy's). */
typedef struct {
    int64_t stripe_bytes;       /* 0: 512 MiB.  A power of two >= 64 MiB (below 512 MiB a chunk no longer lies in ONE class) */
    int64_t budget_bytes;       /* 0: 200 GiB.  Most physical memory one search may hold while it looks for the classes */
    int64_t limit_bytes;        /* 0: a third of the device memory.  The arena never maps more than this */
    double  search_seconds;     /* 0: 3 s.  A search settles for the classes it has after this long; after a third of it for two
                                 * classes in balance (none gives more than three fifths of the stripes) */
} ginsim_placed_options;

typedef struct {
    int32_t available;          /* 1: the arena exists and spans at least two classes */
    int32_t classes;            /* classes of physical memory found so far (3 on MI355X) */
    int32_t searches;           /* chunk searches so far (one per growth) */
    int32_t failed;             /* 1: a search ended with fewer than two classes: placed requests now return GINSIM_ERR_PLACED */
    int64_t stripe_bytes;
    int64_t mapped_bytes;       /* stripes mapped into the arena */
    int64_t used_bytes;         /* handed out */
    int64_t limit_bytes;
    int64_t stripes_of_class[3];
    int64_t chunks_created;     /* over all searches */
    int64_t chunks_ambiguous;   /* chunks no class claimed clearly (given back) */
    int64_t probes;             /* timed fills */
    int64_t peak_held_bytes;    /* most physical memory a search held at once */
    double  search_seconds;     /* all searches together */
    double  last_search_seconds;
    double  anchor_ms;          /* time of a conflict-free pair fill of two stripes (calibrated on single streams) */
    char    stripe_classes[256];/* class letter of the first 255 stripes, in address order ("ABCBCA...") */
} ginsim_placed_info;

int  ginsim_placed_configure(ginsim_ctx* ctx, const ginsim_placed_options* options);   /* before the device's arena exists */
int  ginsim_placed_reserve(ginsim_ctx* ctx, size_t bytes);   /* grow the arena so that `bytes` more can be carved from it: ONE search
                                                               * for the regions of a job instead of one per region */
int  ginsim_malloc_placed(ginsim_ctx* ctx, size_t bytes, void** dptr);
int  ginsim_placed_release(ginsim_ctx* ctx);                  /* give the arena's memory back if nothing is carved from it */
int  ginsim_placed_info_get(ginsim_ctx* ctx, ginsim_placed_info* out);
int  ginsim_memcpy_h2d(ginsim_ctx* ctx, void* dst, const void* src, size_t bytes);
int  ginsim_memcpy_d2h(ginsim_ctx* ctx, void* dst, const void* src, size_t bytes);
int  ginsim_memset(ginsim_ctx* ctx, void* dptr, int value, size_t bytes);
/* Page-locked host memory for callers of the host-buffer entry points (ginsim_free_integration, ginsim_memcpy_*): copies from
 * and to such buffers run at the link rate instead of through the driver's staging of pageable memory.  The reference has
 * no counterpart (its arrays are plain NumPy); the Python layer wraps it as ginsim.pinned_empty(). */
int  ginsim_host_alloc(ginsim_ctx* ctx, size_t bytes, void** hptr);
int  ginsim_host_free(ginsim_ctx* ctx /* may be NULL: the pages may outlive the context */, void* hptr);
int  ginsim_sync(ginsim_ctx* ctx);
/* HIP-event timer on the context's stream (the stream every kernel of this context is launched on). */
int  ginsim_timer_begin(ginsim_ctx* ctx);
int  ginsim_timer_end(ginsim_ctx* ctx, float* elapsed_ms);
/* Event pool for timing many launches without synchronising in between: record event `slot` on the
 * context's stream; elapsed(a,b) synchronises on b and returns the time between the two records. */
#define GINSIM_MAX_EVENTS 8192
int  ginsim_event_record(ginsim_ctx* ctx, int32_t slot);
int  ginsim_event_elapsed(ginsim_ctx* ctx, int32_t slot_a, int32_t slot_b, float* elapsed_ms);

/* ---- truth: pathgen.path_gen (gnss_ins_sim/pathgen/pathgen.py:26-329, osr == 1) ------------- */
typedef struct {
    double  ini_pva[9];     /* lat, lon [rad], alt [m], body vel [m/s], yaw pitch roll [rad]  (ins_sim.py:597-610) */
    double  mobility[3];    /* max accel, max angular accel [rad/s^2], max angular rate [rad/s] (ins_sim.py:25) */
    double  fs;             /* IMU / simulation rate [Hz] */
    double  fs_gps;         /* GPS rate [Hz] (used when enable_gps) */
    int32_t ref_frame;      /* 0 NED/LLA, 1 virtual inertial frame */
    int32_t enable_gps;
    int32_t n_seg;          /* rows of motion_def */
    int32_t enable_mag;     /* emit the true magnetic field in the body frame (pathgen.py:273-279) */
    double  geo_mag_n[3];   /* geomagnetic field at the initial position in the N frame [uT] (pathgen.py:164-168; the WMM
                             * evaluation itself is outside the hot path).  ref_frame 1 drops the declination (:169-171). */
} ginsim_pathgen_params;

/* Upper bound of the IMU sample count: sum of ceil(duration*fs) (pathgen.py:116-127). */
int ginsim_pathgen_capacity(const ginsim_pathgen_params* p, const double* motion_def /*[n_seg][9]*/, int64_t* cap);
/* Host buffers, row-major: imu [cap][7] = idx,acc3,gyro3; nav [cap][10] = idx,pos3,velNED3,euler3;
 * gps [cap][8] = idx,pos3,vel3,visibility (may be NULL); odo [cap][5] = idx,dist,vel_b3 (may be NULL).
 * motion_def is not modified (the reference overwrites column 7, pathgen.py:122). */
int ginsim_pathgen(const ginsim_pathgen_params* p, const double* motion_def, int64_t cap,
                   double* imu, double* nav, double* gps, double* odo, double* mag /*[cap][4] = idx,mag3 or NULL*/,
                   int64_t* n_out, int64_t* m_out);

/* ---- ABI 6: the leaves of the path under their own entry points (host code, for hosted plugins and tools that call the
 * reference functions of the same names one state at a time; ginsim_pathgen and the kernels run the same code inline) ---- */
/* pathgen.calc_true_sensor_output (pathgen.py:331-411): c_nb [3][3] row-major (body -> nav, as the reference passes it); g is
 * used in ref_frame 1 only.  Outputs: acc, gyro (body frame), vel_dot_n, pos_dot_n (lat/lon/alt rates in ref_frame 0). */
int ginsim_calc_true_sensor_output(const double* pos_n, const double* vel_b, const double* att, const double* c_nb,
                                   const double* vel_dot_b, const double* att_dot, int32_t ref_frame, double g,
                                   double* acc, double* gyro, double* vel_dot_n, double* pos_dot_n);
/* pathgen.parse_motion_def (pathgen.py:413-439): seg[0] = command type 1..5, seg[1..3] attitude, seg[4..6] velocity command. */
int ginsim_parse_motion_def(const double* seg, const double* att, const double* vel, double* att_com, double* vel_com);
/* attitude.euler_update_zyx (attitude.py:679-721): x = [yaw, pitch, roll], w body rate, one step of dt. */
int ginsim_euler_update_zyx(const double* x, const double* w, double dt, double* y);

/* ---- Monte-Carlo fused kernel: noise injection + mechanisation + end-point error ------------- */
#define GINSIM_ALGO_FREE 1   /* demo_algorithms/free_integration.py:63-174      */
#define GINSIM_ALGO_ODO  2   /* demo_algorithms/free_integration_odo.py:63-160  */

typedef struct {            /* one 3-axis sensor: pathgen.acc_gen / gyro_gen / bias_drift (pathgen.py:441-594) */
    double  bias[3];        /* err['b'] */
    double  gm_a[3];        /* 1 - 1/(fs*tau)                        (pathgen.py:583); 0 when tau is inf */
    double  gm_b[3];        /* drift*sqrt(1-exp(-2/(fs*tau)))        (pathgen.py:586); drift when tau is inf */
    double  white[3];       /* rw / sqrt(dt)                         (pathgen.py:496, 558) */
    int32_t white_drift[3]; /* 1: tau is inf -> drift[j] = gm_b*N[j] (pathgen.py:593) */
    int32_t reserved;
} ginsim_sensor_model;

/* ABI 5.  The vibration term of one 3-axis sensor (pathgen.py:476-492 accel, :538-556 gyro; Sim.__parse_env, ins_sim.py:642-701,
 * turns the env strings into these numbers).  meas = truth + bias + drift + white + VIB, added last as the reference does:
 *   type 1 'random'      vib[j][k] = amp[k] N[j][k]                     (:485-488, :547-550; three more normals per sample)
 *   type 2 'sinusoidal'  vib[j][k] = amp[k] sin(omega_dt j + phase[k])  omega_dt = 2 pi freq dt; phase = 0 for the accelerometer
 *                        (:489-492), one uniform draw per run and axis times 2 pi for the gyroscope (:551-555)
 *   type 3 'psd' (ABI 8) vib[j][k] = series[k][j mod period][run]       (:479-484, :541-546: time_series_from_psd.py:16-63, a
 *                        random-phase inverse FFT of `period` <= 16384 points per run and axis, tiled to n); the series are made
 *                        on the device by ginsim_vib_psd_series BEFORE the launch and read by it
 * Vibration launches run on the general-sensor-model lane-per-run kernels (ginsim_mc_variant 0; fp64 and fp32 -- the fp32 term
 * is defined operation by operation, csrc/mc_kernel_f32.hip add_vibration) or, sensors only for few runs, on the time-parallel
 * series kernels (variant 2); given sensors refuse it.  Type 3 runs on the fp64 lane-per-run kernels only. */
#define GINSIM_VIB_NONE       0
#define GINSIM_VIB_RANDOM     1
#define GINSIM_VIB_SINUSOIDAL 2
#define GINSIM_VIB_PSD        3
typedef struct {
    int32_t type;           /* GINSIM_VIB_* */
    int32_t random_phase;   /* sinusoidal: 1 = a uniform phase per run and axis (gyro_gen), 0 = phase 0 (acc_gen) */
    double  amp[3];         /* vib_def['x'/'y'/'z']: 1 sigma (random) or peak (sinusoidal), m/s^2 or rad/s */
    double  omega_dt;       /* sinusoidal: ((2.0 * pi) * freq) * dt, rounded as the reference's left-to-right product */
    /* ---- ABI 8: type 3 ---- */
    const double* series;   /* device [3][period][runs], run fastest: what ginsim_vib_psd_series wrote for THESE runs */
    int64_t period;         /* N of time_series_from_psd.py:36-43: n (n + 1 when n is odd), at most 16384 */
} ginsim_vibration;

/* ABI 8.  The vibration series of one 3-axis sensor from a single-sided power spectral density, for `runs` Monte-Carlo runs at once
 * (pathgen.py:479-484 / :541-546 call time_series_from_psd.time_series_from_psd(sxx, freq, fs, n) per run and axis):
 *   x = real(ifft(X)),  X[k] = a[k] exp(i pi z[k]) for k = 0 .. period/2,  Hermitian above  (time_series_from_psd.py:51-57)
 * amp (HOST, [3][period / 2 + 1]) = a = sqrt(sxx' period fs), sxx' the PSD interpolated to the period's frequency grid with its
 * interior bins halved (:44-50) -- leaf arithmetic of O(period), left to the caller (ginsim.psd_amplitudes in the Python layer).
 * z[k] (time_series_from_psd.py:52, np.random.randn(L)) are the normals the 'random' vibration of the same sensor would draw at
 * SAMPLE k (streams 10 / 11 accel, 12 / 13 gyro of run run_offset + r): x, y, z of bin k.  One batched complex-to-real FFT
 * (hipFFT, loaded at run time) per block of runs, then a transposition into out (DEVICE, [3][period][runs], run fastest), scaled
 * by 1 / period.  sensor: 0 accelerometer, 1 gyroscope.
 * halve_per_run = 1 reproduces what the reference does when the PSD is GIVEN on the period's own grid (freq.shape[0] ==
 * period / 2 + 1): it then halves the caller's array in place at every call (:49 without the copy np.interp makes), so run r of a
 * Sim.run() sees interior bins scaled by 0.5^(r + 1); amp is then a of the array as given (nothing halved). */
int ginsim_vib_psd_series(ginsim_ctx* ctx, const double* amp, int64_t period, int64_t runs, uint64_t run_offset, uint64_t seed,
                          int32_t sensor, int32_t halve_per_run, double* out);

typedef struct {
    int64_t  n;             /* IMU samples per run */
    int64_t  runs;          /* Monte-Carlo runs on this device */
    uint64_t run_offset;    /* global id of this device's first run (enters the RNG counter) */
    uint64_t seed;          /* Philox key */
    double   fs;
    int32_t  ref_frame;     /* 0 NED/LLA (free_integration.py:117-172), 1 virtual inertial (:83-116) */
    int32_t  algo_mask;     /* GINSIM_ALGO_* bits; both algorithms see the same sensor realisation; 0 = sensors only */
    int32_t  earth_rot;     /* FreeIntegration(earth_rot=...) (free_integration.py:19, 150-152) */
    int32_t  n_ini;         /* columns of the initial-state table (free_integration.py:42-61) */
    uint64_t ini_first;     /* value of FreeIntegration.run_times before this batch (free_integration.py:69, 85-87) */
    int32_t  ini_has_g;     /* 10th element = externally supplied gravity (free_integration.py:59-61) */
    int32_t  given_sensors; /* 1: read gyro/accel[/odo] from in_* instead of generating them */
    ginsim_sensor_model accel, gyro;
    double   odo_scale, odo_stdv;          /* pathgen.odo_gen (pathgen.py:627-641) */
    double   ref_end[9];    /* truth att(3) pos(3) vel(3) at the last sample, for the end-point error */
    /* device pointers */
    const double* ini;        /* [n_ini][10] */
    const double* ref_accel;  /* [n][3] truth specific force (pathgen imu cols 1-3) */
    const double* ref_gyro;   /* [n][3] truth angular rate   (pathgen imu cols 4-6) */
    const double* ref_odo;    /* [n]    truth forward speed  (pathgen odo col 2), NULL if unused */
    const double* in_accel;   /* given_sensors: [3][n][runs] */
    const double* in_gyro;    /* given_sensors: [3][n][runs] */
    const double* in_odo;     /* given_sensors: [n][runs], NULL if unused */
    double* out_accel;        /* [3][n][runs] or NULL  (dmgr.accel, ins_sim.py:491-493) */
    double* out_gyro;         /* [3][n][runs] or NULL  (dmgr.gyro,  ins_sim.py:494-496) */
    double* out_odo;          /* [n][runs]    or NULL  (dmgr.odo,   ins_sim.py:504-506) */
    double* out_traj[2];      /* per algorithm bit: [9][n][runs] = att3,pos3,vel3 or NULL */
    double* out_end[2];       /* per algorithm bit: [9][runs] end-point error (att wrapped to [-pi,pi]) or NULL */
    /* tuning / telemetry (0 / NULL = defaults) */
    uint64_t* wave_trace;     /* device [n_waves][4] or NULL: HW_ID, XCC_ID, s_memtime at wave start and end */
    int32_t   block_threads;  /* workgroup size: 0 (default), 64, 128 or 256 */
    int32_t   end_pos_ned;    /* ref_frame 0: report the end-point position error in local NED metres (extra_opt='ned',
                               * ins_data_manager.py:474-488, 542-552) instead of [rad, rad, m] */
    int32_t   precision;      /* 0: fp64 (default).  1: fp32 kernel -- out_accel/out_gyro/out_odo/out_traj point to FLOAT
                               * buffers of the same [component][sample][run] shape; the position planes of out_traj hold
                               * the displacement from the run's initial position (ECEF-based for ref_frame 1, LLA for 0);
                               * out_end stays double; in_* (given_sensors) stay double and are rounded to float as read.
                               * The fp32 kernel consumes the SAME normals as the fp64 one (they are single-precision numbers)
                               * and every operation of it is one defined IEEE operation: oracle/c/ginsim_oracle.c
                               * (oracle_mc_run_f32) reproduces its series to the bit. */
    int32_t   proc_pos_ned;   /* process statistics of the position error in local NED metres (ref_frame 0 only) */
    /* ---- statistics WITHOUT trajectories (ABI 2): what Sim.results() needs when the series are not materialised ---- */
    const double* ref_nav;    /* [n][9] truth att3, pos3, vel3 of every sample; needed when out_proc is set */
    int64_t   proc_first;     /* first sample of the process-error window (err_stats_start, ins_data_manager.py:771-781) */
    double*   out_proc[2];    /* per algorithm bit: [3][9][runs] = max|e|, mean, std(ddof=0) of the error over samples
                               * >= proc_first, accumulated online in the fused kernel (InsDataMgr.__process_error_stats,
                               * ins_data_manager.py:761-795, on array_error :519-553), or NULL.  ONE algorithm bit per launch,
                               * generate mode, fp64. */
    double*   out_end_ned[2]; /* per algorithm bit: [9][runs] end-point error with the position error in local NED metres
                               * (extra_opt='ned', ins_data_manager.py:542-552) next to out_end, or NULL (ref_frame 0 only) */
    /* ---- ABI 4 ---- */
    int32_t   sensor_layout;  /* layout of out_accel / out_gyro / out_odo.  0 (default): [axis][sample][run], run fastest -- what
                               * the lane-per-run kernels write.  1: SERIES-major [run][axis][sample] ([run][sample] for the
                               * odometer) -- every series contiguous, the input layout of ginsim_allan (series_stride = n) and of
                               * the reference's own per-run arrays dmgr.accel.data[i] (ins_sim.py:491-496).  Only sensors-only
                               * launches (algo_mask 0) take it; with <= 1024 runs and >= 2048 samples they run on the
                               * time-parallel series kernels (ginsim_mc_variant reports 2), otherwise layout 1 is refused.
                               * PERFORMANCE CLIFF: with layout 0 and 2..1024 runs a sensors-only launch takes the lane-per-run
                               * kernel, whose time loop is ONE sequential chain over n per lane -- for long series (config 5:
                               * 1 440 000 samples) seconds instead of a millisecond.  A caller that wants few long series
                               * sets sensor_layout = 1 (or asks ginsim_mc_variant first: 2 = time-parallel). */
    int32_t   proc_plain_sums; /* online process statistics, the variants whose sums are shifted per LAUNCH (ref_frame 0 free integration; see
                               * csrc/mc_kernel.hip, Proc): 0 (default) = the sums are kept about the error of the first run's initial
                               * state against the truth at sample 0, so an error that is nearly constant over the window keeps its true
                               * std (np.std, ins_data_manager.py:761-795).  1 = the caller states that every run starts ON the truth
                               * (that error is zero): the sums are taken as they are -- what nine zero shifts give, nine
                               * double-precision subtractions per step fewer (BASELINE config 3: 2.3 %).  Stating it of runs that
                               * start off the truth brings back the rounding floor of ~1.5e-8 |mean| on the std.  (Was reserved4,
                               * which had to be 0.) */
    /* ---- ABI 5: vibration, Sim(env={'acc': ..., 'gyro': ...}) -> the vib term of pathgen.acc_gen / gyro_gen ---- */
    ginsim_vibration vib_accel, vib_gyro;
} ginsim_mc_params;

int ginsim_mc_run(ginsim_ctx* ctx, const ginsim_mc_params* p);

/* Which kernel ginsim_mc_run would launch for these parameters (telemetry for profiles; results are bit-identical between
 * 0 and 1): 0 = one wavefront per 64 runs does noise + mechanisation (mc_kernel / mc_kernel_f32), 1 = wave-specialised
 * producer/consumer workgroups (mc_kernel_split / mc_kernel_f32_split), chosen for batches of <= 1024 wavefronts and,
 * for one algorithm in ref_frame 1 (two producer groups: three wavefronts per SIMD), at every size; 2 (ABI 4) = the
 * time-parallel series kernels (sensors only, <= 1024 runs, >= 2048 samples, sensor_layout 1 or one run): a lane is a
 * SAMPLE, the Gauss-Markov recurrence a weighted scan -- same normals, sums associated differently (an ulp of the terms). */
int ginsim_mc_variant(const ginsim_mc_params* p, int32_t* variant);
/* ABI 3: the NAME of that kernel as rocprofv3 reports it, without arguments (e.g. "ginsim::mc_kernel_split<1, 1, false, 2,
 * true>"), written by the same dispatch code that launches it -- profiles and bench.py attribute to what really runs. */
int ginsim_mc_kernel_name(const ginsim_mc_params* p, char* buf, size_t cap);

/* ---- auxiliary sensors of a Monte-Carlo batch: pathgen.gps_gen (pathgen.py:596-625) and pathgen.mag_gen (:643-661).
 *      FreeIntegration does not consume them, so they are generated only when they are to be kept. */
typedef struct {
    int64_t  n;             /* IMU samples (magnetometer rate) */
    int64_t  m;             /* GPS samples */
    int64_t  runs;
    uint64_t run_offset;
    uint64_t seed;
    double   gps_sigma[6];  /* position sigma (already in rad,rad,m for ref_frame 0: pathgen.py:616-619) and velocity sigma */
    double   mag_si[9];     /* soft-iron matrix, row major */
    double   mag_hi[3];     /* hard iron [uT] */
    double   mag_std[3];    /* noise sigma [uT] */
    const double* ref_gps;  /* device [m][6] or NULL */
    const double* ref_mag;  /* device [n][3] or NULL */
    double*  out_gps;       /* device [6][m][runs] or NULL */
    double*  out_mag;       /* device [3][n][runs] or NULL */
} ginsim_aux_params;

int ginsim_aux_sensors(ginsim_ctx* ctx, const ginsim_aux_params* p);

/* ---- end-point statistics: InsDataMgr.__end_point_error_stats / __array_stats
 *      (gnss_ins_sim/sim/ins_data_manager.py:717-759, 797-808) ------------------------------------ */
typedef struct {
    double count;
    double mean[9];
    double m2[9];           /* sum of squared deviations from mean; std(ddof=0) = sqrt(m2/count) */
    double maxabs[9];
} ginsim_stats;             /* 28 doubles; mergeable across devices (Chan et al.) */

int ginsim_end_stats(ginsim_ctx* ctx, const double* end_err /*device [9][runs]*/, int64_t runs, ginsim_stats* host_out);

/* The same reduction without blocking the host: _begin enqueues the reduction and the copy of the record into pinned
 * host slot `slot` (0..7) on the context's stream; _finish waits for that slot only and returns the record.  A caller
 * can enqueue the next ginsim_mc_run between the two, so the host-side merge / multi-GPU exchange of batch k overlaps
 * the integration of batch k+1. */
int ginsim_end_stats_begin(ginsim_ctx* ctx, const double* end_err, int64_t runs, int32_t slot);
int ginsim_end_stats_finish(ginsim_ctx* ctx, int32_t slot, ginsim_stats* host_out);

/* ---- multi-GPU: Monte-Carlo runs shard over ranks by global run id (run_offset), one process and one context per GPU; the
 *      ONE exchange of the path is every rank's statistics record to every rank.  ABI 3 puts it behind the
 *      boundary: an RCCL all-gather enqueued on the context's stream right behind the on-device reduction (librccl is
 *      resolved at run time on first use; single-GPU callers never load it).  The reference has no counterpart
 *      (gnss_ins_sim/sim/ins_sim.py:490 is a serial loop).
 *      Bootstrap: rank 0 calls ginsim_comm_unique_id and hands the 128 bytes to the other ranks through whatever launcher the
 *      caller uses (an environment variable, a file, torch.distributed's store ...); every rank then calls ginsim_comm_init. */
#define GINSIM_COMM_ID_BYTES 128
/* ABI 4: can librccl be reached from this process (dlopen + dlsym, nothing else)?  Every rank calls it and the verdicts are
 * reduced BEFORE anyone enters the collective ginsim_comm_init: a rank that failed alone would leave the others waiting there. */
int ginsim_comm_probe(void);
/* rank 0 only (it starts RCCL's bootstrap listener for this id) */
int ginsim_comm_unique_id(unsigned char* id /*[GINSIM_COMM_ID_BYTES]*/);
/* Collective over the nranks ranks (ncclCommInitRank): returns an error on every rank or on none once all have entered; a rank
 * that never enters (it failed earlier) leaves the others blocked inside RCCL -- hence ginsim_comm_probe.  On failure the
 * context is left without a communicator and may retry. */
int ginsim_comm_init(ginsim_ctx* ctx, int32_t nranks, int32_t rank, const unsigned char* id);
int ginsim_comm_destroy(ginsim_ctx* ctx);
/* ABI 7: what the communicator ITSELF answers (ncclCommCount, ncclCommUserRank, ncclCommCuDevice) -- evidence that RCCL saw the
 * ranks the caller meant, for the benchmark line (`rccl_ranks`); -1 where the loaded librccl lacks a query. */
int ginsim_comm_query(ginsim_ctx* ctx, int32_t* nranks, int32_t* rank, int32_t* device);
/* ginsim_end_stats_begin / _finish over ALL ranks: reduction of this rank's end errors (runs may be 0: an empty record) ->
 * all-gather of the 28-double records -> copy into pinned slot `slot` (0..7), all on the context's stream; _finish waits for
 * that slot only and returns the Chan merge of the non-empty records in rank order (the same on every rank). */
int ginsim_end_stats_all_begin(ginsim_ctx* ctx, const double* end_err, int64_t runs, int32_t slot);
int ginsim_end_stats_all_finish(ginsim_ctx* ctx, int32_t slot, ginsim_stats* merged);

/* Process-error statistics of every run: InsDataMgr.__process_error_stats (ins_data_manager.py:761-795) over
 * array_error (:519-553): e[j] = traj[j] - ref[j] for samples j >= first_sample, attitude wrapped to [-pi,pi];
 * pos_ned != 0 (ref_frame 0, extra_opt='ned'): LLA error -> metres in the local NED frame of the reference (:542-552).
 * traj: device [9][n][runs]; ref: device [n][9] (att3,pos3,vel3 truth).  host_out [runs][3][9] = max|e|, mean, std(ddof=0). */
int ginsim_process_stats(ginsim_ctx* ctx, const double* traj, const double* ref, int64_t n, int64_t runs,
                         int64_t first_sample, int32_t pos_ned, double* host_out);
/* End-point statistics recomputed from kept trajectories (e.g. with pos_ned after the run): last-sample error of
 * every run on the device, then the same reduction as ginsim_end_stats. */
int ginsim_end_stats_from_traj(ginsim_ctx* ctx, const double* traj, const double* ref, int64_t n, int64_t runs,
                               int32_t pos_ned, ginsim_stats* host_out);
/* ABI 4: the same two statistics over the FLOAT trajectories of the fp32 kernel (precision 1), whose position planes hold the
 * displacement from the run's initial position: origin = device [n_ini][3], the initial positions (ECEF metres for ref_frame
 * 1, LLA for ref_frame 0) of the initial-state table the launch used, ini_first as in ginsim_mc_params.  The position of a sample
 * is formed as origin + displacement in fp64 and everything after it is the fp64 arithmetic of the calls above. */
int ginsim_process_stats_f32(ginsim_ctx* ctx, const float* traj, const double* ref, int64_t n, int64_t runs, int64_t first_sample,
                             int32_t pos_ned, const double* origin, int32_t n_ini, uint64_t ini_first, double* host_out);
int ginsim_end_stats_from_traj_f32(ginsim_ctx* ctx, const float* traj, const double* ref, int64_t n, int64_t runs, int32_t pos_ned,
                                   const double* origin, int32_t n_ini, uint64_t ini_first, ginsim_stats* host_out);
int ginsim_stats_merge(const ginsim_stats* parts, int32_t nparts, ginsim_stats* out);

/* ---- data access: pull selected runs out of a [ncomp][n][runs] device series into host [nsel][n][ncomp] */
int ginsim_gather_runs(ginsim_ctx* ctx, const double* series, int32_t ncomp, int64_t n, int64_t runs,
                       const int64_t* run_ids /*host*/, int32_t nsel, double* host_out);
/* same for a SERIES-major sensor buffer (sensor_layout 1): series [runs][ncomp][n] -> host [nsel][n][ncomp] */
int ginsim_gather_series(ginsim_ctx* ctx, const double* series, int32_t ncomp, int64_t n, int64_t runs,
                         const int64_t* run_ids /*host*/, int32_t nsel, double* host_out);
/* same for a float series written by the fp32 kernel (values widened to double on the way out) */
int ginsim_gather_runs_f32(ginsim_ctx* ctx, const float* series, int32_t ncomp, int64_t n, int64_t runs,
                           const int64_t* run_ids /*host*/, int32_t nsel, double* host_out);

/* ---- given-data mechanisation with host buffers: the plugin's .run(set_of_input) boundary
 *      (free_integration.py:63-174 / free_integration_odo.py:63-160).  gyro/accel [R][n][3], odo [R][n],
 *      ini [n_ini][10]; outputs att/pos/vel [R][n][3].  algo is ONE GINSIM_ALGO_* bit. */
int ginsim_free_integration(ginsim_ctx* ctx, int32_t algo, int32_t ref_frame, double fs, int32_t earth_rot,
                            const double* gyro, const double* accel, const double* odo, int64_t R, int64_t n,
                            const double* ini, int32_t n_ini, int32_t ini_has_g, uint64_t ini_first,
                            double* att, double* pos, double* vel);

/* ---- Allan variance: allan.allan_var (gnss_ins_sim/allan/allan.py:18-59) for a batch of series, as the Allan
 *      plugin applies it to each sensor axis (demo_algorithms/allan_analysis.py:33-49).
 *      x: device pointer, series s occupies x[s*series_stride .. +n).  Outputs (host): tau[cap] and
 *      avar[nseries][cap] (Allan VARIANCE; the plugin reports its square root); *ntau = number of averaging
 *      factors (0 when the series is shorter than 9 s, allan.py:30-31). */
int ginsim_allan(ginsim_ctx* ctx, const double* x, int64_t n, int32_t nseries, int64_t series_stride, double fs,
                 double* tau, double* avar, int32_t* ntau, int32_t cap);

/* Device-to-device re-layout of a Monte-Carlo series [ncomp][n][runs] (run fastest, what ginsim_mc_run writes) into
 * per-run contiguous series [runs][ncomp][n] -- the input layout of ginsim_allan, so that the Allan plugin
 * (demo_algorithms/allan_analysis.py:33-49) works on the generated sensors without a host round trip. */
int ginsim_runs_to_series(ginsim_ctx* ctx, const double* series, int32_t ncomp, int64_t n, int64_t runs, double* out);

/* ---- RNG self-test hook: first `count` normal pairs of (seed, run, stream) computed ON DEVICE.  Stream s at sample j is
 *      half (s & 1) of the Philox4x32-7 block with counter (j, s >> 1, run_lo, run_hi) and key = seed; each of its two words
 *      gives one normal by a piecewise-cubic inversion of the tail probability that is defined operation by operation in
 *      IEEE single precision on a committed coefficient table (csrc/fastmath.hpp normal_icdf; spelled out in
 *      oracle/philox.py, which reproduces the device's normals to the bit).  |z| <= 6.23.  The normals are single-precision
 *      numbers returned as doubles.  host_words, if given, receives the raw block with counter (j, stream, run_lo, run_hi). ---- */
int ginsim_rng_normals(ginsim_ctx* ctx, uint64_t seed, uint64_t run, uint32_t stream, int64_t count,
                       double* host_z0, double* host_z1, uint32_t* host_words /*[count][4] or NULL*/);

/* Test hook: the device's normal transform applied to caller-chosen Philox words (z0 from w[i][0], z1 from w[i][1]; w[i][2..3]
 * unused), so that corner cases no seed will produce in a test (smallest / largest magnitude, every segment edge of the
 * coefficient table) can be pinned against the oracle. */
int ginsim_normal_transform(ginsim_ctx* ctx, const uint32_t* host_words /*[count][4]*/, int64_t count, double* host_z0,
                            double* host_z1);

#ifdef __cplusplus
}
#endif
#endif /* GINSIM_H */
