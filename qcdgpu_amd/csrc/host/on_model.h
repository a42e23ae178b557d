// on_model.h -- host-side run API of the O(N) scalar model, shaped like the reference's O(N) tree:
// model_CL::model with MODEL_ON = 1 (O(N)/suncl/suncl.h:47-437) driven by O(N)/QCDGPU.cpp:75-241.
// Same run parameters under the same names, the same init-file keys (O(N)/suncl/suncl.cpp:249-319),
// the same call sequence and result-file layout.  Every device action goes through the qcd_on_*
// C-ABI of include/qcdgpu_amd.h; there is no CPU path here.
#pragma once

#include <ctime>
#include <string>
#include <vector>

#include "model.h"

namespace model_HIP {

// indices of model::Analysis used by the O(N) model (O(N)/suncl/suncl.h:226-242)
#define DM_ON_S_total 1
#define DM_ON_Plq_total 2        // "Field_variance"
#define DM_ON_S_spat 3
#define DM_ON_S_temp 4
#define DM_ON_Plq_spat 5         // "Field"
#define DM_ON_Plq_temp 6         // "Field^2"
#define DM_ON_Acc_rate_even 11
#define DM_ON_Acc_rate_odd 12
#define DM_ON_Acc_rate_total 13
#define DM_ON_Correlator1 15
#define DM_ON_Correlator2 16
#define DM_ON_max 16

class on_model {
public:
    // run parameters, names as in the reference (O(N)/suncl/suncl.h:67-138)
    int desired_device = 0;
    bool device_select = false;
    int lattice_nd = 4;
    int lattice_group = 1;
    int lattice_type = 1;
    int lattice_full_size[4] = {18, 18, 18, 4};
    model::enum_model_precision precision = model::model_precision_double;
    model::enum_model_starts ints = model::model_start_cold;
    int INIT = 1;
    int NAV = 50;
    int ITER = 200;
    int NITER = 10;
    int NHIT = 10;
    unsigned int ON_SK_group = 4;
    double ON_z = 0.1561573990981998;
    double ON_lambda = 0.25;
    double ON_zeta = 0.25;
    double ON_eta = 0.1;
    double ON_b = 9.661840260273483;
    double ON_max_U = 0.0;            // derived by the library (O(N)/suncl/suncl.cpp:1246-1253)
    unsigned int correlator_X = 4, correlator_Y = 4, correlator_Z = 4, correlator_T = 4;
    bool get_acceptance_rate = true;
    bool get_correlators = true;
    bool get_actions_avr = true;
    bool get_plaquettes_avr = true;
    bool turnoff_updates = false;
    bool profiling = false;
    std::string version = "1.0";
    std::string path = "./";
    std::string finishpath = "./";
    std::string fprefix = "o1-";

    PRNG* PRNG0;
    data_analysis Analysis[DM_ON_max + 1];
    std::string header;
    char* timestart = nullptr;
    char* timeend = nullptr;
    time_t ltimestart = 0, ltimeend = 0;
    std::string result_file;
    double sweep_ms = 0.0;            // device time of all sweeps (profiling)
    long sweeps_done = 0;

    on_model();
    ~on_model();

    void lattice_get_init_file(char* file);
    void lattice_init(void);
    void lattice_simulate(void);
    void lattice_analysis(void);
    void lattice_write_results(void);
    void lattice_print_measurements(void);
    const char* lattice_make_header(void);
    const char* lattice_make_header2(void);

private:
    qcd_on_ctx* ctx = nullptr;
    std::vector<double> s_, f1_, f2_, c1_, c2_, are_, aro_;
    void lattice_update(void);
    void lattice_measure(int index);
    void check(int rc, const char* what);
};

}  // namespace model_HIP
