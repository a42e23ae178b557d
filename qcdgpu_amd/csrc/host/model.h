// model.h -- host-side run API of the SU(N) sweep engine, shaped like the reference's
// model_CL::model (SU(N)/suncl/suncl.h:61-467): the same public run parameters under the same
// names, the same call sequence
//
//     lattice_get_init_file -> lattice_init -> lattice_simulate -> lattice_analysis
//                           -> lattice_write_results -> lattice_print_measurements
//
// (SU(N)/QCDGPU.cpp:237-259), the same init.dat keys and the same result-file layout.  Everything
// that touches a device goes through the C-ABI in include/qcdgpu_amd.h; there is no CPU path here.
//
// Differences that are deliberate:
//   * the number of BigLattice parts is a run-time property (length of the N1 = {..} vector), not
//     the compile-time NPARTS of the reference (SU(N)/Makefile:19);
//   * a failing device call throws model_error instead of calling exit(-1) inside the service
//     layer (SU(N)/clinterface/clinterface.cpp:225-284); main() turns it into exit status 1.
#pragma once

#include <cstdio>
#include <ctime>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/qcdgpu_amd.h"

namespace model_HIP {

// indices of model::Analysis (SU(N)/suncl/suncl.h:224-255); only the observables of this engine
#define DM_Wilson_loop 0
#define DM_S_total 1
#define DM_Plq_total 2
#define DM_S_spat 3
#define DM_S_temp 4
#define DM_Plq_spat 5
#define DM_Plq_temp 6
#define DM_Polyakov_loop 7
#define DM_Polyakov_loop_im 8
#define DM_Polyakov_loop_P2 9
#define DM_Polyakov_loop_P4 10
// F_mu_nu block (SU(N)/suncl/suncl.h:235-252): three planes x (re, im) for the first and the second
// Gell-Mann index, then the four |F| columns
#define DM_Fmunu_3 11
#define DM_Fmunu_8 17
#define DM_Fmunu_abs_3_re 23
#define DM_Fmunu_abs_3_im 24
#define DM_Fmunu_abs_8_re 25
#define DM_Fmunu_abs_8_im 26
#define DM_max 26

struct model_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// one "KEY = value" line of init.dat (GPU_init_parameters, SU(N)/clinterface/clinterface.h)
struct init_parameter {
    std::string Variable;
    std::string txtVarVal;
    int iVarVal = 0;
    double fVarVal = 0.0;
    std::vector<int> ivVarVal;  // "{a, b, ...}" vectors of the BigLattice keys
};
std::vector<init_parameter> get_init_file(const char* finitf);

// the slice of PRNG_CL::PRNG this path uses (SU(N)/random/random.h)
class PRNG {
public:
    enum PRNG_generators { PRNG_generator_RANLUX0, PRNG_generator_RANLUX1, PRNG_generator_RANLUX2,
                           PRNG_generator_RANLUX3, PRNG_generator_RANLUX4 };
    PRNG_generators PRNG_generator = PRNG_generator_RANLUX3;
    unsigned int PRNG_randseries = 0;  // 0: taken from the system time at initialisation
    unsigned int PRNG_srandtime = 0;
    void parameters_setup(const char* Variable, int iVarVal, const char* txtVarVal);
    int print_generator(char* header, int header_size) const;
    int lux_level() const { return (int)PRNG_generator; }
};

// per-series statistics (analysis_CL::analysis::data_analysis, SU(N)/data_analysis/data_analysis.h)
struct data_analysis {
    const char* data_name = nullptr;
    std::vector<double> data;
    double mean_value = 0.0;
    double variance = 0.0;
    double GPU_last_value = 0.0;
};
// mean and variance over samples 1..n-1 (SU(N)/data_analysis/data_analysis.cpp:107-135)
void lattice_data_analysis(data_analysis* d);

// one part of a BigLattice (SubLattice, SU(N)/suncl/suncl.h:470-699)
struct SubLattice {
    int Nx = 0, Ny = 0, Nz = 0, Nt = 0;
    int desired_platform = 0;
    int desired_device = 0;
    qcd_ctx* ctx = nullptr;
};

class model {
public:
    enum enum_model_starts { model_start_hot, model_start_cold, model_start_gid };
    enum enum_model_precision { model_precision_single, model_precision_double, model_precision_mixed };

    // run parameters, names as in the reference (SU(N)/suncl/suncl.h:76-167)
    int desired_platform = 0;
    int desired_device = 0;
    bool device_select = false;
    int lattice_nd = 4;
    int lattice_group = 3;
    int lattice_full_size[4] = {32, 32, 32, 32};
    int lattice_domain_size[4] = {32, 32, 32, 32};
    enum_model_precision precision = model_precision_double;
    // model::convert_precision_to_uint (SU(N)/suncl/suncl.cpp:732-737) and the bytes of one stored real
    unsigned int precision_word() const { return precision == model_precision_double ? 2u : (precision == model_precision_mixed ? 3u : 1u); }
    size_t real_bytes() const { return precision == model_precision_double ? 8 : 4; }
    int INIT = 1;
    int NAV = 0;
    int ITER = 2;
    int NITER = 1;
    int NHIT = 10;
    int NHITPar = 1;
    int NOVERRELAX = 0;   // overrelaxation sweeps after every heat-bath sweep (not in the reference)
    bool CHB2 = false;    // BigLattice working cycles with one edge exchange per sweep (the reference's -D CHB2 build)
    int RNGSHARE = 0;     // sites of one parity per RANLUX stream (1, 2, 4; 0: chosen by the library from the lattice size)
    double BETA = 6.5;
    double PHI = 0.0;
    double OMEGA = 0.0;
    enum_model_starts ints = model_start_cold;
    int PL_level = 2;
    int wilson_R = 7;
    int wilson_T = 2;
    std::string version = "1.5";
    std::string path = "./";
    std::string finishpath = "./";
    std::string fprefix = "su3-";
    bool turnoff_updates = false;
    bool turnoff_gramschmidt = false;
    bool get_actions_avr = true;
    bool get_plaquettes_avr = true;
    bool get_wilson_loop = false;  // GETWILSON: the wilson_R x wilson_T loop as a further column
    bool get_Fmunu = false;        // FMUNU: F_mu_nu traces of the planes xy, xz, yz ("H field")
    bool get_F0mu = false;         // F0MU: of the planes xt, yt, zt ("E field")
    int Fmunu_index1 = 3;          // Gell-Mann index of the first trace (FMUNU1, FMUNU2, FMUNU4; default 3)
    int Fmunu_index2 = 8;          // of the second (FMUNU5, FMUNU6, FMUNU7; default 8)
    bool get_actions_diff = false; // S(x), S(y), S(z): the action plane by plane (suncl.h:144; init key GETSDIFF is an addition)
    bool profiling = false;        // GPU_debug.profiling
    // lattice state files (SU(N)/suncl/suncl.h:80,154-175; defaults of main, QCDGPU.cpp:203,215)
    bool turnoff_config_save = true;                 // do not write configurations
    double write_lattice_state_every_secs = 15.0 * 60.0;
    std::string fstate;                              // state file to continue from when INIT = 0
    unsigned int PRNG_counter = 0;                   // update passes done (the reference: produce() calls)
    int NAV_counter = 0;
    int ITER_counter = 0;
    std::string state_file;                          // name of the file lattice_save_state wrote last
    int stop_at_sample = -1;                         // leave lattice_simulate once this many samples exist (an interrupted run)
    bool stopped_early = false;

    PRNG* PRNG0;

    // BigLattice: lattice_Nparts > 1 cuts the lattice into x-slabs (N1 = {..}); parts are mapped
    // to devices by DEVICE = {..} (several parts may share a device)
    int lattice_Nparts = 1;
    std::vector<SubLattice> SubLat;
    // the parts as a ring below the C-ABI: a pass (update of every part + UpdateEdges) is one call.
    // Transport: RCCL when every part has a device of its own, copies otherwise (TRANSPORT = COPY | RCCL overrides)
    qcd_ring* ring = nullptr;
    int ring_transport = -1;          // -1: choose; QCD_RING_COPY; QCD_RING_RCCL
    std::string ring_transport_name;  // what the run uses, for the header

    data_analysis Analysis[DM_max + 1];

    std::string header;
    char* timestart = nullptr;
    char* timeend = nullptr;
    time_t ltimestart = 0, ltimeend = 0;
    std::string result_file;  // name of the file lattice_write_results wrote

    model();
    ~model();

    void lattice_get_init_file(char* file);
    void lattice_init(void);
    void lattice_simulate(void);
    void lattice_analysis(void);
    void lattice_write_results(void);
    void lattice_print_measurements(void);
    const char* lattice_make_header(void);
    const char* lattice_make_header2(void);
    void print_time_detailed(void);
    // <path><prefix>YY-Mon-DD-HH-MM-SS.qcg (SU(N)/suncl/suncl.cpp:1857-2079)
    void lattice_save_state(void);
    void lattice_load_state(void);

    // halo exchange after a pass over links of direction dir on the sites of one parity
    // (model::UpdateEdges, SU(N)/suncl/suncl.cpp:4005-4061)
    void UpdateEdges(int dir, int parity);

private:
    std::vector<double> plq_[3], act_[3], pl_[4];
    std::vector<double> wl_;
    std::vector<double> fmunu_[12];                // per sample: the twelve F_mu_nu averages
    std::string fmunu_names_[16];
    // plane-resolved series, [axis][plane * ITER + sample]: S_total_diff and (PL, PL_im) diff
    std::vector<double> s_diff_[3], pl_diff_[3], pl_diff_im_[3];
    void measure_profiles(int index);
    void write_profiles(FILE* stream);
    std::vector<double> plq_raw_[2], pl_raw_[4];   // per sample: the lattice sums the table entries derive from
    void derive_sample(int index);
    std::string device_names_;
    void sweep(bool working = false);   // working: a sweep of the working cycles (the CHB2 variant applies there only)
    void measure(int index);
    void check(int rc, const char* what);
    unsigned int LOAD_state = 0;
    int rng_share_used_ = 1;   // sites per random stream the contexts were created with (RNGSHARE, or the library's choice)
    std::vector<unsigned int> lattice_make_bin_header(void) const;
    bool lattice_load_bin_header(const unsigned int* head);
    unsigned int energies_size() const;
    unsigned int energies_size_F() const;
    unsigned int table_row_size() const;
    time_t last_save_ = 0;
    unsigned int state_sizes_[4] = {0, 0, 0, 0};   // section lengths read from a state file's header
};

}  // namespace model_HIP
