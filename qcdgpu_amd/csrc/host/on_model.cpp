// on_model.cpp -- see on_model.h.  Host logic only; the device work is behind the qcd_on_* C-ABI.
#include "on_model.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace model_HIP {

on_model::on_model()
{
    PRNG0 = new PRNG();
    PRNG0->PRNG_randseries = 7;   // O(N)/QCDGPU.cpp:113
    static const char* names[DM_ON_max + 1] = {"", "S_total", "Field_variance", "S_spat", "S_temp", "Field", "Field^2",
                                               "", "", "", "", "AR_even", "AR_odd", "AR_total", "", "Correlator",
                                               "Correlator(+1)"};   // O(N)/suncl/suncl.cpp:708-903
    for (int i = 0; i <= DM_ON_max; i++) Analysis[i].data_name = names[i];
}

on_model::~on_model()
{
    if (ctx) qcd_on_destroy(ctx);
    delete PRNG0;
    free(timestart);
    free(timeend);
}

void on_model::check(int rc, const char* what)
{
    if (rc != QCD_OK) throw model_error(std::string(what) + ": " + qcd_last_error());
}

// parameters_setup, O(N)/suncl/suncl.cpp:249-319 (keys of observables this model does not have are
// accepted and ignored, as there); PRECISION (1 single, 2 double: convert_uint_to_precision,
// suncl.cpp:428-433) is an addition -- the reference fixes the precision in its main()
void on_model::lattice_get_init_file(char* file)
{
    std::vector<init_parameter> ps = get_init_file(file);
    for (const init_parameter& p : ps) {
        const char* k = p.Variable.c_str();
        if (!strcmp(k, "DEVICE")) { desired_device = p.iVarVal; device_select = true; }
        if (!strcmp(k, "GROUP")) lattice_group = p.iVarVal;
        if (!strcmp(k, "TYPE")) lattice_type = p.iVarVal;
        if (!strcmp(k, "ND")) lattice_nd = p.iVarVal;
        if (!strcmp(k, "L1") || !strcmp(k, "N1")) lattice_full_size[0] = p.iVarVal;
        if (!strcmp(k, "L2") || !strcmp(k, "N2")) lattice_full_size[1] = p.iVarVal;
        if (!strcmp(k, "L3") || !strcmp(k, "N3")) lattice_full_size[2] = p.iVarVal;
        if (!strcmp(k, "L4") || !strcmp(k, "N4")) lattice_full_size[3] = p.iVarVal;
        if (!strcmp(k, "LS") || !strcmp(k, "NS")) lattice_full_size[0] = lattice_full_size[1] = lattice_full_size[2] = p.iVarVal;
        if (!strcmp(k, "LT") || !strcmp(k, "NT")) lattice_full_size[3] = p.iVarVal;
        if (!strcmp(k, "SKGROUP")) ON_SK_group = (unsigned int)p.iVarVal;
        if (!strcmp(k, "ITER")) ITER = p.iVarVal;
        if (!strcmp(k, "NITER")) NITER = p.iVarVal;
        if (!strcmp(k, "NHIT")) NHIT = p.iVarVal;
        if (!strcmp(k, "NAV")) NAV = p.iVarVal;
        if (!strcmp(k, "ON_Z")) ON_z = p.fVarVal;
        if (!strcmp(k, "ON_ZETA")) ON_zeta = p.fVarVal;
        if (!strcmp(k, "ON_ETA")) ON_eta = p.fVarVal;
        if (!strcmp(k, "ON_B")) ON_b = p.fVarVal;
        if (!strcmp(k, "ON_LAMBDA")) ON_lambda = p.fVarVal;
        if (!strcmp(k, "RANDSERIES")) PRNG0->PRNG_randseries = (unsigned int)p.iVarVal;
        PRNG0->parameters_setup(k, p.iVarVal, p.txtVarVal.c_str());
        if (!strcmp(k, "INTS")) ints = p.iVarVal == 1 ? model::model_start_cold : model::model_start_hot;   // convert_uint_to_start, :422-427
        if (!strcmp(k, "OUTPUTPATH")) path = p.txtVarVal;
        if (!strcmp(k, "FINISHPATH")) finishpath = p.txtVarVal;
        if (!strcmp(k, "GETRETRACE")) get_plaquettes_avr = true;
        if (!strcmp(k, "PRECISION")) precision = p.iVarVal == 2 ? model::model_precision_double : model::model_precision_single;
    }
}

static char* current_datetime(time_t* t)
{
    time(t);
    char* result = (char*)calloc(256, sizeof(char));
    snprintf(result, 256, "%s", ctime(t));
    return result;
}

// lattice_make_header + model_make_header, O(N)/suncl/suncl.cpp:332-368, 483-491
const char* on_model::lattice_make_header(void)
{
    char buf[8192];
    int j = 0;
    const int n = (int)sizeof(buf);
    char name[256] = "?";
    qcd_device_name(desired_device, name, sizeof(name));
    j += snprintf(buf + j, n - j, " GPU O(%u) simulator (Skalozub action for O(%u)) %s\n\n", lattice_group, ON_SK_group, version.c_str());
    j += snprintf(buf + j, n - j, " Monte Carlo simulation of %uD O(%u) LGT\n\n", lattice_nd, lattice_group);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    j += snprintf(buf + j, n - j, " Active OpenCL platform   : %s\n", "AMD HIP (ROCm)");
    j += snprintf(buf + j, n - j, " Active OpenCL device     : %s\n", name);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    j += snprintf(buf + j, n - j, " lattice size                : %3u x %3u x %3u x %3u\n", lattice_full_size[0], lattice_full_size[1],
                  lattice_full_size[2], lattice_full_size[3]);
    j += snprintf(buf + j, n - j, " init                        : %i\n", INIT);
    j += snprintf(buf + j, n - j, " ints (0=hot, 1=cold)        : %i\n", ints == model::model_start_hot ? 0 : 1);
    j += PRNG0->print_generator(buf + j, n - j);
    j += snprintf(buf + j, n - j, " nav                         : %i\n", NAV);
    j += snprintf(buf + j, n - j, " niter                       : %i\n", NITER);
    j += snprintf(buf + j, n - j, " iter (# of samples)         : %i\n", ITER);
    j += snprintf(buf + j, n - j, " nhit                        : %i\n", NHIT);
    j += snprintf(buf + j, n - j, " precision                   : %s\n", precision == model::model_precision_double ? "double" : "single");
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    j += snprintf(buf + j, n - j, " eta                         : %16.13e\n", ON_eta);
    j += snprintf(buf + j, n - j, " z                           : %16.13e\n", ON_z);
    j += snprintf(buf + j, n - j, " b                           : %16.13e\n", ON_b);
    j += snprintf(buf + j, n - j, " lambda                      : %16.13e\n", ON_lambda);
    j += snprintf(buf + j, n - j, " zeta                        : %16.13e\n", ON_zeta);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    header = buf;
    return header.c_str();
}

// lattice_init + model_lattice_init, O(N)/suncl/suncl.cpp:369-401, 1241-1445
void on_model::lattice_init(void)
{
    if (INIT == 0) throw model_error("continuing a saved O(N) run (INIT = 0) is not implemented");
    if (lattice_nd != 4) throw model_error("ND must be 4");
    if (lattice_group != 1 || lattice_type != 1) throw model_error("the O(N) program runs GROUP = 1, TYPE = 1 (one real number per site)");
    if (precision == model::model_precision_mixed) throw model_error("mixed precision is not implemented");
    PRNG0->PRNG_srandtime = (unsigned int)time(NULL);
    if (!PRNG0->PRNG_randseries) PRNG0->PRNG_randseries = PRNG0->PRNG_srandtime;
    else PRNG0->PRNG_srandtime = PRNG0->PRNG_randseries;

    qcd_on_params p;
    memset(&p, 0, sizeof(p));
    for (int i = 0; i < 4; i++) p.L[i] = lattice_full_size[i];
    p.precision = precision == model::model_precision_double ? 1 : 0;
    p.nhit = NHIT;
    p.sk_group = (int)ON_SK_group;
    p.z = ON_z;
    p.lambda = ON_lambda;
    p.zeta = ON_zeta;
    p.eta = ON_eta;
    p.b = ON_b;
    p.start_mode = ints == model::model_start_hot ? QCD_START_HOT : QCD_START_COLD;
    p.lux_level = PRNG0->lux_level();
    p.rand_series = PRNG0->PRNG_randseries;
    p.device = desired_device;
    check(qcd_on_create(&ctx, &p), "qcd_on_create");
    check(qcd_on_max_u(ctx, &ON_max_U), "qcd_on_max_u");
    lattice_make_header();
}

// lattice_update, O(N)/suncl/suncl.cpp:2228-2256: odd pass, then even pass
void on_model::lattice_update(void)
{
    if (turnoff_updates) return;
    if (profiling) {
        double ms = 0.0;
        check(qcd_on_timed_sweeps(ctx, 1, &ms), "qcd_on_timed_sweeps");
        sweep_ms += ms;
    } else {
        check(qcd_on_sweep(ctx, 1), "qcd_on_sweep");
    }
    sweeps_done++;
}

// lattice_measure, O(N)/suncl/suncl.cpp:2278-2289
void on_model::lattice_measure(int index)
{
    const int step[4] = {(int)correlator_X, (int)correlator_Y, (int)correlator_Z, (int)correlator_T};
    double out[5];
    check(qcd_on_measure(ctx, step, out), "qcd_on_measure");
    s_[index] = out[0];
    f1_[index] = out[1];
    f2_[index] = out[2];
    c1_[index] = out[3];
    c2_[index] = out[4];
    unsigned long long cnt[3];
    check(qcd_on_counters(ctx, cnt), "qcd_on_counters");
    // denominator of the reference: lattice_full_site * NHIT * NITER (suncl.cpp:884, 895) over the
    // sums of the LAST sweep (its buffer is cleared in front of every sweep, suncl.cpp:2229)
    const double sites = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
    const double den = sites * NHIT * NITER;
    are_[index] = sweeps_done ? (double)cnt[1] / den : 0.0;
    aro_[index] = sweeps_done ? (double)cnt[2] / den : 0.0;
}

// lattice_simulate, O(N)/suncl/suncl.cpp:2092-2156: sample 0 on the initial field, NAV
// thermalisation sweeps, then ITER-1 times { NITER sweeps ; measure }
void on_model::lattice_simulate(void)
{
    if (!ctx) throw model_error("lattice_simulate before lattice_init");
    if (ITER < 1) throw model_error("ITER < 1");
    for (std::vector<double>* v : {&s_, &f1_, &f2_, &c1_, &c2_, &are_, &aro_}) v->assign(ITER, 0.0);
    free(timestart);
    timestart = current_datetime(&ltimestart);
    check(qcd_on_init_lattice(ctx), "qcd_on_init_lattice");
    int ITER_counter = 0;
    lattice_measure(ITER_counter++);
    for (int i = 0; i < NAV; i++) {
        lattice_update();
        if (i % 10 == 0) printf("\rGPU thermalization [%i]", i);
    }
    for (int i = ITER_counter; i < ITER; i++) {
        for (int j = 0; j < NITER; j++) lattice_update();
        if (i % 10 == 0) printf("\rGPU working iteration [%u]", i);
        lattice_measure(ITER_counter++);
    }
    free(timeend);
    timeend = current_datetime(&ltimeend);
    printf("\rGPU simulations are done\n");
}

// lattice_analysis, O(N)/suncl/suncl.cpp:681-780, 879-906, and the joint / difference series of
// O(N)/data_analysis/data_analysis.cpp:139-234
void on_model::lattice_analysis(void)
{
    Analysis[DM_ON_S_spat].data = s_;
    Analysis[DM_ON_S_temp].data.assign(ITER, 0.0);     // the .y half of the action buffer is never written
    Analysis[DM_ON_S_total].data = s_;                 // 0.5 * (S_spat + S_spat)
    Analysis[DM_ON_Plq_spat].data = f1_;
    Analysis[DM_ON_Plq_temp].data = f2_;
    Analysis[DM_ON_Plq_total].data.assign(ITER, 0.0);
    Analysis[DM_ON_Correlator1].data.assign(ITER, 0.0);
    Analysis[DM_ON_Correlator2].data.assign(ITER, 0.0);
    Analysis[DM_ON_Acc_rate_even].data = are_;
    Analysis[DM_ON_Acc_rate_odd].data = aro_;
    Analysis[DM_ON_Acc_rate_total].data.assign(ITER, 0.0);
    for (int i = 0; i < ITER; i++) {
        Analysis[DM_ON_Plq_total].data[i] = f2_[i] - f1_[i] * f1_[i];
        // lattice_data_analysis_sub: the connected part, when the field is measured too
        Analysis[DM_ON_Correlator1].data[i] = c1_[i] - (get_plaquettes_avr ? f1_[i] * f1_[i] : 0.0);
        Analysis[DM_ON_Correlator2].data[i] = c2_[i] - (get_plaquettes_avr ? f1_[i] * f1_[i] : 0.0);
        Analysis[DM_ON_Acc_rate_total].data[i] = 0.5 * (are_[i] + aro_[i]);
    }
    for (int i = 1; i <= DM_ON_max; i++)
        if (!Analysis[i].data.empty()) lattice_data_analysis(&Analysis[i]);
    // lattice_data_analysis_joint_variance takes the mean from the means (data_analysis.cpp:227)
    Analysis[DM_ON_Plq_total].mean_value = Analysis[DM_ON_Plq_temp].mean_value -
                                           Analysis[DM_ON_Plq_spat].mean_value * Analysis[DM_ON_Plq_spat].mean_value;
}

// lattice_make_header2, O(N)/suncl/suncl.cpp:508-631 (no CPU re-measurement lines: this program has no CPU path)
const char* on_model::lattice_make_header2(void)
{
    char buf[8192];
    int j = 0;
    const int n = (int)sizeof(buf);
    const double dif = difftime(ltimeend, ltimestart);
    j += snprintf(buf + j, n - j, " Start time               : %s", timestart ? timestart : "\n");
    j += snprintf(buf + j, n - j, " Finish time              : %s", timeend ? timeend : "\n");
    j += snprintf(buf + j, n - j, " Elapsed time             : %i:%2.2i:%2.2i:%2.2i\n", (int)(dif / 24 / 3600), (((int)dif / 3600) % 3600),
                  (((int)dif / 60) % 60), (((int)dif) % 60));
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_ON_S_total].data_name, Analysis[DM_ON_S_total].mean_value);
    j += snprintf(buf + j, n - j, " Variance %-16s: % 16.13e\n", Analysis[DM_ON_S_total].data_name, Analysis[DM_ON_S_total].variance);
    j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_ON_Acc_rate_total].data_name, Analysis[DM_ON_Acc_rate_total].mean_value);
    j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_ON_Plq_spat].data_name, Analysis[DM_ON_Plq_spat].mean_value);
    j += snprintf(buf + j, n - j, " Variance %-16s: % 16.13e\n", Analysis[DM_ON_Plq_total].data_name, Analysis[DM_ON_Plq_total].mean_value);
    j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_ON_Correlator1].data_name, Analysis[DM_ON_Correlator1].mean_value);
    j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_ON_Correlator2].data_name, Analysis[DM_ON_Correlator2].mean_value);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    for (int idx : {DM_ON_S_total, DM_ON_Plq_spat, DM_ON_Correlator1, DM_ON_Correlator2})
        j += snprintf(buf + j, n - j, " GPU last %-16s: % 16.13e\n", Analysis[idx].data_name, Analysis[idx].GPU_last_value);
    j += snprintf(buf + j, n - j, " Data fields:\n");
    j += snprintf(buf + j, n - j, "    #, ");
    std::vector<int> cols;
    if (get_plaquettes_avr) cols.insert(cols.end(), {DM_ON_Plq_spat, DM_ON_Plq_temp, DM_ON_Plq_total});
    if (get_actions_avr) cols.insert(cols.end(), {DM_ON_S_spat, DM_ON_S_temp, DM_ON_S_total});
    if (get_correlators) cols.insert(cols.end(), {DM_ON_Correlator1, DM_ON_Correlator2});
    if (get_acceptance_rate) cols.insert(cols.end(), {DM_ON_Acc_rate_even, DM_ON_Acc_rate_odd, DM_ON_Acc_rate_total});
    for (int c : cols) j += snprintf(buf + j, n - j, "%-22s", Analysis[c].data_name);
    j += snprintf(buf + j, n - j, "\n");
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    header += buf;
    return header.c_str();
}

// lattice_write_results, O(N)/suncl/suncl.cpp:909-970: <path><prefix>YY-Mon-DD-HH-MM-SS.txt
void on_model::lattice_write_results(void)
{
    const char* header2 = lattice_make_header2();
    printf("%s\n", header2);
    char name[64] = "";
    if (timeend && strlen(timeend) >= 24)
        snprintf(name, sizeof(name), "%.2s-%.3s-%.2s-%.2s-%.2s-%.2s.txt", timeend + 22, timeend + 4, timeend + 8, timeend + 11,
                 timeend + 14, timeend + 17);
    result_file = path + fprefix + name;
    FILE* stream = fopen(result_file.c_str(), "w+");
    if (!stream) throw model_error("cannot write " + result_file);
    fputs(header.c_str(), stream);
    std::vector<int> cols;
    if (get_plaquettes_avr) cols.insert(cols.end(), {DM_ON_Plq_spat, DM_ON_Plq_temp, DM_ON_Plq_total});
    if (get_actions_avr) cols.insert(cols.end(), {DM_ON_S_spat, DM_ON_S_temp, DM_ON_S_total});
    if (get_correlators) cols.insert(cols.end(), {DM_ON_Correlator1, DM_ON_Correlator2});
    if (get_acceptance_rate) cols.insert(cols.end(), {DM_ON_Acc_rate_even, DM_ON_Acc_rate_odd, DM_ON_Acc_rate_total});
    for (int i = 0; i < ITER; i++) {
        fprintf(stream, "%5i", i);
        for (int c : cols) fprintf(stream, " % 16.13e", Analysis[c].data[i]);
        fprintf(stream, "\n");
    }
    if (fclose(stream)) printf("The file was not closed!\n");
}

// lattice_print_measurements, O(N)/suncl/suncl.cpp:632-680: first five and the last sample
void on_model::lattice_print_measurements(void)
{
    printf("\n #        ");
    if (get_plaquettes_avr) printf(" %-11s %-11s", Analysis[DM_ON_Plq_spat].data_name, Analysis[DM_ON_Plq_temp].data_name);
    if (get_actions_avr) printf(" %-11s %-11s", Analysis[DM_ON_S_spat].data_name, Analysis[DM_ON_S_temp].data_name);
    if (get_acceptance_rate) printf(" %-11s %-11s", Analysis[DM_ON_Acc_rate_even].data_name, Analysis[DM_ON_Acc_rate_odd].data_name);
    printf("\n");
    for (int i = 0; i < ITER; i++) {
        if ((i < 5) || (i == (ITER - 1))) {
            printf("[%5u]: ", i);
            if (get_plaquettes_avr) printf(" % 10.8f % 10.8f", Analysis[DM_ON_Plq_spat].data[i], Analysis[DM_ON_Plq_temp].data[i]);
            if (get_actions_avr) printf(" % 10.8f % 10.8f", Analysis[DM_ON_S_spat].data[i], Analysis[DM_ON_S_temp].data[i]);
            if (get_acceptance_rate) printf(" % 10.8f % 10.8f", Analysis[DM_ON_Acc_rate_even].data[i], Analysis[DM_ON_Acc_rate_odd].data[i]);
            printf("\n");
        }
    }
    if (profiling && sweeps_done)
        printf("\n sweeps %ld, %.3f ms per sweep on the device\n", sweeps_done, sweep_ms / sweeps_done);
}

}  // namespace model_HIP
