// model.cpp -- host-side run API over the C-ABI (see model.h).
#include "model.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <strings.h>

namespace model_HIP {

// ---------------------------------------------------------------------------------------------
// init.dat tokenizer: "KEY = value" per line, leading blanks and tabs skipped, text after '#'
// ignored, the value read as int, as double and as raw text; "{a, b, ...}" integer vectors for
// the BigLattice keys (SU(N)/clinterface/clinterface.cpp:2060-2154, SU(N)/suncpp/IO/io.cpp:42-116)

std::vector<init_parameter> get_init_file(const char* finitf)
{
    std::vector<init_parameter> result;
    FILE* stream = fopen(finitf, "r");
    if (!stream) return result;
    char line[250];
    while (fgets(line, sizeof(line), stream) != NULL) {
        const size_t len = strlen(line);
        const size_t eq = strcspn(line, "=");
        if (eq >= len) continue;  // no '=': not a parameter line
        size_t s1 = 0;
        while (line[s1] == ' ' || line[s1] == '\t') s1++;
        long e1 = (long)eq - 1;
        while (e1 >= 0 && (line[e1] == ' ' || line[e1] == '\t')) e1--;
        init_parameter p;
        if (e1 >= (long)s1) p.Variable.assign(line + s1, (size_t)(e1 - (long)s1 + 1));
        long fin = (long)strcspn(line, "#") - 1;
        while (fin >= 0 && (line[fin] == ' ' || line[fin] == '\t' || line[fin] == '\n' || line[fin] == '\r')) fin--;
        size_t s3 = eq + 1;
        while (line[s3] == ' ' || line[s3] == '\t') s3++;
        if (fin >= (long)s3) p.txtVarVal.assign(line + s3, (size_t)(fin - (long)s3 + 1));
        sscanf(p.txtVarVal.c_str(), "%d", &p.iVarVal);
        sscanf(p.txtVarVal.c_str(), "%lf", &p.fVarVal);
        if (!p.txtVarVal.empty() && p.txtVarVal[0] == '{') {
            const char* t = p.txtVarVal.c_str() + 1;
            int v, n;
            while (sscanf(t, " %d%n", &v, &n) == 1) {
                p.ivVarVal.push_back(v);
                t += n;
                while (*t == ' ' || *t == ',') t++;
            }
        }
        result.push_back(p);
    }
    fclose(stream);
    return result;
}

// ---------------------------------------------------------------------------------------------
// PRNG (SU(N)/random/random.cpp:140-172)

void PRNG::parameters_setup(const char* parameter, int ivalue, const char* text_value)
{
    if (!strcmp(parameter, "RANDSERIES")) PRNG_randseries = (unsigned)ivalue;
    if (!strcmp(parameter, "PRNG")) {
        if (!strcmp(text_value, "RANLUX0")) PRNG_generator = PRNG_generator_RANLUX0;
        if (!strcmp(text_value, "RANLUX1")) PRNG_generator = PRNG_generator_RANLUX1;
        if (!strcmp(text_value, "RANLUX2")) PRNG_generator = PRNG_generator_RANLUX2;
        if (!strcmp(text_value, "RANLUX3")) PRNG_generator = PRNG_generator_RANLUX3;
        if (!strcmp(text_value, "RANLUX4")) PRNG_generator = PRNG_generator_RANLUX4;
        // the reference's other generators are not compiled into its kernels either
        // (SU(N)/random/random.cl:5-12); RANLUX3 stays selected for them
    }
}

int PRNG::print_generator(char* header, int header_size) const
{
    int j = 0;
    j += snprintf(header + j, header_size - j, " random series               : %u\n", PRNG_srandtime);
    j += snprintf(header + j, header_size - j, " PRN generator               : RANLUX%d\n", (int)PRNG_generator);
    return j;
}

// SU(N)/data_analysis/data_analysis.cpp:107-135: sample 0 (the initial configuration) is excluded
void lattice_data_analysis(data_analysis* d)
{
    const int last_index = (int)d->data.size() - 1;
    d->mean_value = 0.0;
    d->variance = 0.0;
    if (last_index < 0) return;
    for (int i = 1; i <= last_index; i++) d->mean_value += d->data[i];
    if (last_index > 0) d->mean_value /= (double)last_index;
    for (int i = 1; i <= last_index; i++) d->variance += pow(d->data[i] - d->mean_value, 2);
    if (last_index > 0) d->variance /= (double)last_index;
    d->GPU_last_value = d->data[last_index];
}

// ---------------------------------------------------------------------------------------------

model::model()
{
    PRNG0 = new PRNG();
    // defaults of main() (SU(N)/QCDGPU.cpp:98-220)
    PRNG0->PRNG_generator = PRNG::PRNG_generator_RANLUX3;
    PRNG0->PRNG_randseries = 54;
    static const char* names[DM_max + 1] = {"Wilson_loop", "S_total", "Plq_total", "S_spat", "S_temp",
                                            "Plq_spat", "Plq_temp", "Polyakov_loop", "Polyakov_loop_im",
                                            "Polyakov_loop_P2", "Polyakov_loop_P4"};
    for (int i = 0; i <= DM_max; i++) Analysis[i].data_name = names[i];
}

model::~model()
{
    qcd_ring_destroy(ring);
    for (auto& s : SubLat) qcd_destroy(s.ctx);
    delete PRNG0;
    free(timestart);
    free(timeend);
}

void model::check(int rc, const char* what)
{
    if (rc != QCD_OK) throw model_error(std::string(what) + ": " + qcd_last_error());
}

// key table of SU(N)/suncl/suncl.cpp:305-459 (the keys of observables this engine does not
// measure are accepted and ignored, like unknown keys are in the reference)
void model::lattice_get_init_file(char* file)
{
    std::vector<init_parameter> parameters = get_init_file(file);
    std::vector<int> n1, devices, rnds;
    for (const init_parameter& p : parameters) {
        const char* v = p.Variable.c_str();
        if (!strcmp(v, "PLATFORM")) desired_platform = p.iVarVal;
        if (!strcmp(v, "DEVICE")) {
            if (!p.ivVarVal.empty()) devices = p.ivVarVal;
            else desired_device = p.iVarVal;
            device_select = true;
        }
        if (!strcmp(v, "GROUP")) {
            lattice_group = p.iVarVal;
            char buf[16];
            snprintf(buf, sizeof(buf), "su%i-", lattice_group);
            fprefix = buf;
        }
        if (!strcmp(v, "ND")) lattice_nd = p.iVarVal;
        if (!strcmp(v, "L1")) lattice_full_size[0] = lattice_domain_size[0] = p.iVarVal;
        if (!strcmp(v, "L2")) lattice_full_size[1] = lattice_domain_size[1] = p.iVarVal;
        if (!strcmp(v, "L3")) lattice_full_size[2] = lattice_domain_size[2] = p.iVarVal;
        if (!strcmp(v, "L4")) lattice_full_size[3] = lattice_domain_size[3] = p.iVarVal;
        if (!strcmp(v, "LS"))
            for (int i = 0; i < 3; i++) lattice_full_size[i] = lattice_domain_size[i] = p.iVarVal;
        if (!strcmp(v, "LT") && lattice_nd >= 1 && lattice_nd <= 4)  // ND must precede LT, as in the reference
            lattice_full_size[lattice_nd - 1] = lattice_domain_size[lattice_nd - 1] = p.iVarVal;
        if (!strcmp(v, "N1") && !p.ivVarVal.empty()) n1 = p.ivVarVal;
        if (!strcmp(v, "ITER")) ITER = p.iVarVal;
        if (!strcmp(v, "NITER")) NITER = p.iVarVal;
        if (!strcmp(v, "NHIT")) NHIT = p.iVarVal;
        if (!strcmp(v, "BETA")) BETA = p.fVarVal;
        if (!strcmp(v, "PHI")) PHI = p.fVarVal;
        if (!strcmp(v, "OMEGA")) OMEGA = p.fVarVal;
        if (!strcmp(v, "NAV")) NAV = p.iVarVal;
        if (!strcmp(v, "NOVERRELAX")) NOVERRELAX = p.iVarVal;   // extension: the reference has no overrelaxation
        if (!strcmp(v, "CHB2")) CHB2 = p.iVarVal != 0;          // the reference's build flag -D CHB2 (SU(N)/Makefile:5-20) as a key
        if (!strcmp(v, "RNGSHARE")) RNGSHARE = p.iVarVal;       // extension: sites per RANLUX stream (qcd_params::rng_share)
        if (!strcmp(v, "INTS"))
            ints = (p.iVarVal == 0) ? model_start_hot : (p.iVarVal == 2 ? model_start_gid : model_start_cold);
        if (!strcmp(v, "RANDSERIES") && !p.ivVarVal.empty()) rnds = p.ivVarVal;
        PRNG0->parameters_setup(v, p.ivVarVal.empty() ? p.iVarVal : p.ivVarVal[0], p.txtVarVal.c_str());
        if (!strcmp(v, "OUTPUTPATH")) path = p.txtVarVal;
        // added key: how the edges of a BigLattice travel (default: RCCL when every part has its own device)
        if (!strcmp(v, "TRANSPORT"))
            ring_transport = !strcasecmp(p.txtVarVal.c_str(), "RCCL") ? QCD_RING_RCCL : QCD_RING_COPY;
        if (!strcmp(v, "FINISHPATH")) finishpath = p.txtVarVal;
        if (!strcmp(v, "GETWILSON")) get_wilson_loop = true;
        if (!strcmp(v, "GETRETRACE")) get_plaquettes_avr = true;
        if (!strcmp(v, "PL_LEVEL")) PL_level = p.iVarVal;
        if (!strcmp(v, "GETSDIFF")) get_actions_diff = true;
        // SU(N)/suncl/suncl.cpp:412-452, 2104-2112
        if (!strcmp(v, "TURNOFFFMUNU")) get_Fmunu = get_F0mu = false;
        if (!strcmp(v, "FMUNU")) { get_Fmunu = true; get_F0mu = false; }
        if (!strcmp(v, "F0MU")) { get_F0mu = true; get_Fmunu = false; }
        if (!strcmp(v, "FMUNU1")) Fmunu_index1 = 1;
        if (!strcmp(v, "FMUNU2")) Fmunu_index1 = 2;
        if (!strcmp(v, "FMUNU4")) Fmunu_index1 = 4;
        if (!strcmp(v, "FMUNU5")) Fmunu_index2 = 5;
        if (!strcmp(v, "FMUNU6")) Fmunu_index2 = 6;
        if (!strcmp(v, "FMUNU7")) Fmunu_index2 = 7;
        if (!strcmp(v, "WILSONR")) wilson_R = p.iVarVal;
        if (!strcmp(v, "WILSONT")) wilson_T = p.iVarVal;
    }
    if (n1.size() > 1) {
        lattice_Nparts = (int)n1.size();
        SubLat.assign(n1.size(), SubLattice());
        for (size_t k = 0; k < n1.size(); k++) {
            SubLat[k].Nx = n1[k];
            SubLat[k].Ny = lattice_full_size[1];
            SubLat[k].Nz = lattice_full_size[2];
            SubLat[k].Nt = lattice_full_size[3];
            SubLat[k].desired_device = k < devices.size() ? devices[k] : desired_device;
        }
    } else if (!devices.empty()) {
        desired_device = devices[0];
    }
}

const char* model::lattice_make_header(void)
{
    // SU(N)/suncl/suncl.cpp:468-578
    char buf[16384];
    int j = 0;
    const int n = (int)sizeof(buf);
    j += snprintf(buf + j, n - j, " GPU SU(%u) simulator %s (QCDGPU-m-Nx-8.1)\n\n", lattice_group, version.c_str());
    j += snprintf(buf + j, n - j, " Monte Carlo simulation of %uD SU(%u) LGT\n\n", lattice_nd, lattice_group);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    if (lattice_Nparts > 1) {
        for (int k = 0; k < lattice_Nparts; k++) {
            char name[256] = "?";
            qcd_device_name(SubLat[k].desired_device, name, sizeof(name));
            j += snprintf(buf + j, n - j, " Part %i:                     \n", k + 1);
            j += snprintf(buf + j, n - j, "    Active OpenCL platform   : %s\n", "AMD HIP (ROCm)");
            j += snprintf(buf + j, n - j, "    Active OpenCL device     : %s\n", name);
        }
    } else {
        char name[256] = "?";
        qcd_device_name(desired_device, name, sizeof(name));
        j += snprintf(buf + j, n - j, " Active OpenCL platform   : %s\n", "AMD HIP (ROCm)");
        j += snprintf(buf + j, n - j, " Active OpenCL device     : %s\n", name);
    }
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    j += snprintf(buf + j, n - j, " lattice size                : %3u x %3u x %3u x %3u\n", lattice_full_size[0],
                  lattice_full_size[1], lattice_full_size[2], lattice_full_size[3]);
    if (lattice_Nparts > 1) {
        j += snprintf(buf + j, n - j, " ---------------------------------------------------\n");
        for (int k = 0; k < lattice_Nparts; k++)
            j += snprintf(buf + j, n - j, " Part %i                      : %3u x %3u x %3u x %3u\n", k + 1, SubLat[k].Nx,
                          SubLat[k].Ny, SubLat[k].Nz, SubLat[k].Nt);
        j += snprintf(buf + j, n - j, " ---------------------------------------------------\n");
    }
    j += snprintf(buf + j, n - j, " init                        : %i\n", INIT);
    j += snprintf(buf + j, n - j, " ints (0=hot, 1=cold, 2=gid) : %i\n",
                  ints == model_start_hot ? 0 : (ints == model_start_cold ? 1 : 2));
    j += PRNG0->print_generator(buf + j, n - j);
    j += snprintf(buf + j, n - j, " nav                         : %i\n", NAV);
    j += snprintf(buf + j, n - j, " niter                       : %i\n", NITER);
    j += snprintf(buf + j, n - j, " iter (# of samples)         : %i\n", ITER);
    j += snprintf(buf + j, n - j, " nhit                        : %i\n", NHIT);
    j += snprintf(buf + j, n - j, " nhitPar                     : %i\n", NHITPar);
    if (precision == model_precision_single) j += snprintf(buf + j, n - j, " precision                   : single\n");
    if (precision == model_precision_mixed) j += snprintf(buf + j, n - j, " precision                   : mixed\n");
    if (precision == model_precision_double) j += snprintf(buf + j, n - j, " precision                   : double\n");
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    // model_make_header, SU(N)/suncl/suncl.cpp:779-795
    j += snprintf(buf + j, n - j, " Wilson loop R               : %i\n", wilson_R);
    j += snprintf(buf + j, n - j, " Wilson loop T               : %i\n", wilson_T);
    if (get_Fmunu) j += snprintf(buf + j, n - j, " FMUNU(%u, %u)\n", Fmunu_index1, Fmunu_index2);   // suncl.cpp:784-787
    if (get_F0mu) j += snprintf(buf + j, n - j, " F0MU(%u, %u)\n", Fmunu_index1, Fmunu_index2);
    j += snprintf(buf + j, n - j, " BETA                        : %16.13e\n", BETA);
    j += snprintf(buf + j, n - j, " PHI   (lambda_3)            : %16.13e\n", PHI);
    j += snprintf(buf + j, n - j, " OMEGA (lambda_8)            : %16.13e\n", OMEGA);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    header = buf;
    return header.c_str();
}

static char* current_datetime(time_t* t)
{
    time(t);
    char* result = (char*)calloc(256, sizeof(char));
    snprintf(result, 256, "%s", ctime(t));
    return result;
}

// lattice_init (SU(N)/suncl/suncl.cpp:618-706): device selection and buffers; one context per part
void model::lattice_init(void)
{
    if (INIT == 0) lattice_load_state();   // run parameters from the state file's header (suncl.cpp:695)
    if (lattice_nd != 4) throw model_error("ND must be 4");
    // the precision is the library's (the reference builds its kernels for one precision, -D PRECISION,
    // SU(N)/suncl/suncl.cpp:3274-3301): qcdgpu is linked with the double library, qcdgpu_f32 with the single one
    if ((qcd_precision() == 2) != (precision == model_precision_double))
        throw model_error(qcd_precision() == 2 ? "this program is linked with the double-precision library: run qcdgpu_f32 for single or mixed precision"
                                               : "this program is linked with the single-precision library: run qcdgpu for double precision");
    // twisted boundary conditions (-D TBC, SU(N)/suncl/suncl.cpp:3286): one uncut lattice, no overrelaxation sweeps
    if ((PHI != 0.0 || OMEGA != 0.0) && (lattice_Nparts > 1 || NOVERRELAX > 0))
        throw model_error("twisted boundary conditions (PHI, OMEGA) are implemented for one uncut lattice and heat-bath sweeps only");
    if ((get_Fmunu || get_F0mu) && lattice_group == 2) {
        // Fmunu_defaults, SU(N)/suncl/suncl.cpp:2094-2101: Pauli matrices, first index 3 or 1, second 2
        Fmunu_index1 = Fmunu_index1 == 1 ? 1 : 3;
        Fmunu_index2 = 2;
    }
    if (get_wilson_loop && lattice_Nparts > 1) throw model_error("the Wilson loop is measured on an uncut lattice only (GETWILSON with N1 = {..})");
    // initialize_CPU (SU(N)/random/random.cpp:174-182): RANDSERIES = 0 means "seed from the clock"
    PRNG0->PRNG_srandtime = (unsigned int)time(NULL);
    if (!PRNG0->PRNG_randseries) PRNG0->PRNG_randseries = PRNG0->PRNG_srandtime;
    else PRNG0->PRNG_srandtime = PRNG0->PRNG_randseries;

    if (lattice_Nparts <= 1) {
        lattice_Nparts = 1;
        SubLat.assign(1, SubLattice());
        SubLat[0].Nx = lattice_full_size[0];
        SubLat[0].Ny = lattice_full_size[1];
        SubLat[0].Nz = lattice_full_size[2];
        SubLat[0].Nt = lattice_full_size[3];
        SubLat[0].desired_device = desired_device;
    }
    int total = 0;
    for (auto& s : SubLat) total += s.Nx;
    if (total != lattice_full_size[0]) throw model_error("N1 = {..} does not add up to L1");

    // sites per random stream: one value for all parts, chosen for the thinnest of them
    int rng_share = RNGSHARE;
    if (rng_share == 0) {
        int min_nx = lattice_full_size[0];
        for (auto& s : SubLat) min_nx = s.Nx < min_nx ? s.Nx : min_nx;
        rng_share = (PHI != 0.0 || OMEGA != 0.0) ? 1 : qcd_rng_share_for(lattice_full_size, min_nx);
    }
    int x0 = 0;
    for (auto& s : SubLat) {
        qcd_params p;
        memset(&p, 0, sizeof(p));
        p.group = lattice_group;
        p.nd = lattice_nd;
        for (int i = 0; i < 4; i++) p.L[i] = lattice_full_size[i];
        p.precision = (int)precision_word();
        p.phi = PHI;
        p.omega = OMEGA;
        p.rng_share = rng_share;
        rng_share_used_ = rng_share;
        p.beta = BETA;
        p.nhit = NHIT;
        p.start_mode = (ints == model_start_hot) ? QCD_START_HOT : (ints == model_start_gid ? QCD_START_GID : QCD_START_COLD);
        p.lux_level = PRNG0->lux_level();
        p.rand_series = PRNG0->PRNG_randseries;
        p.device = s.desired_device;
        p.x0 = x0;
        p.nx_local = s.Nx;
        p.halo = lattice_Nparts > 1 ? 1 : 0;
        check(qcd_create(&s.ctx, &p), "qcd_create");
        check(qcd_set_profiling(s.ctx, profiling ? 1 : 0), "qcd_set_profiling");
        x0 += s.Nx;
    }
    if (lattice_Nparts > 1) {
        // model::UpdateEdges moves the edges through host memory, part after part (suncl.cpp:4005-4061);
        // here the parts form a ring whose passes are issued below the C-ABI, over RCCL when every part
        // has a GPU of its own
        std::vector<qcd_ctx*> parts;
        bool distinct = true;
        for (size_t k = 0; k < SubLat.size(); k++) {
            parts.push_back(SubLat[k].ctx);
            for (size_t j = 0; j < k; j++) distinct = distinct && SubLat[j].desired_device != SubLat[k].desired_device;
        }
        int want = ring_transport >= 0 ? ring_transport : (distinct ? QCD_RING_RCCL : QCD_RING_COPY);
        int rc = qcd_ring_create(&ring, parts.data(), (int)parts.size(), want);
        if (rc != QCD_OK && want == QCD_RING_RCCL && ring_transport < 0) {
            fprintf(stderr, "RCCL transport not available (%s): copying the edges instead\n", qcd_last_error());
            want = QCD_RING_COPY;
            rc = qcd_ring_create(&ring, parts.data(), (int)parts.size(), want);
        }
        check(rc, "qcd_ring_create");
        ring_transport_name = want == QCD_RING_RCCL ? "RCCL" : "copy";
    }
    lattice_make_header();
}

void model::UpdateEdges(int dir, int parity)
{
    if (lattice_Nparts < 2) return;
    for (int k = 0; k < lattice_Nparts; k++) {
        const int k_next = (k + 1) % lattice_Nparts;
        // last owned slice of part k -> left halo of part k+1 (every direction)
        check(qcd_halo_copy(SubLat[k].ctx, 1, SubLat[k_next].ctx, 0, dir, parity), "UpdateEdges");
        // first owned slice of part k+1 -> right halo of part k (nothing reads U_x there)
        if (dir != 0) check(qcd_halo_copy(SubLat[k_next].ctx, 0, SubLat[k].ctx, 1, dir, parity), "UpdateEdges");
    }
}

// one sweep in the reference order: odd X,Y,Z,T then even X,Y,Z,T, edges after every pass
// (SU(N)/suncl/suncl.cpp:5357-5377 and :5026-5138); re-unitarisation is part of the update kernel
void model::sweep(bool working)
{
    if (turnoff_updates) return;
    if (ring && CHB2 && working && NOVERRELAX == 0) {
        // the reference's -D CHB2 build: in the working cycles the edges are copied once per sweep (suncl.cpp:4683-4729)
        check(qcd_ring_sweep_chb2(ring, 1), "qcd_ring_sweep_chb2");
        PRNG_counter += 8;
        return;
    }
    for (int parity = 1; parity >= 0; parity--)
        for (int dir = 0; dir < 4; dir++) {
            if (ring) {
                check(qcd_ring_update(ring, parity, dir), "qcd_ring_update");
            } else {
                for (auto& s : SubLat) check(qcd_update(s.ctx, parity, dir), "qcd_update");
            }
            PRNG_counter++;
        }
    // microcanonical sweeps after the heat-bath sweep (extension, NOVERRELAX = 0 by default)
    for (int n = 0; n < NOVERRELAX; n++)
        for (int parity = 1; parity >= 0; parity--)
            for (int dir = 0; dir < 4; dir++) {
                if (ring) {
                    check(qcd_ring_overrelax(ring, parity, dir), "qcd_ring_overrelax");
                } else {
                    for (auto& s : SubLat) check(qcd_overrelax(s.ctx, parity, dir), "qcd_overrelax");
                }
            }
}

void model::measure(int index)
{
    double plq[2] = {0, 0}, pl[4] = {0, 0, 0, 0};
    for (auto& s : SubLat) {
        double a[2], b[4];
        if (get_plaquettes_avr || get_actions_avr) {
            check(qcd_measure_plaquette_sums(s.ctx, a), "qcd_measure_plaquette_sums");
            plq[0] += a[0];
            plq[1] += a[1];
        }
        if (PL_level > 0) {
            check(qcd_measure_polyakov_sums(s.ctx, b), "qcd_measure_polyakov_sums");
            for (int i = 0; i < 4; i++) pl[i] += b[i];
        }
    }
    for (int i = 0; i < 2; i++) plq_raw_[i][index] = plq[i];
    for (int i = 0; i < 4; i++) pl_raw_[i][index] = pl[i];
    if (get_wilson_loop) check(qcd_measure_wilson(SubLat[0].ctx, wilson_R, wilson_T, &wl_[index]), "qcd_measure_wilson");
    derive_sample(index);
    if (get_actions_diff || PL_level > 2) measure_profiles(index);
    if (get_Fmunu || get_F0mu) {
        // lattice_analysis_Fmunu, SU(N)/suncl/suncl.cpp:1000-1042: site averages over the whole lattice
        double f[12] = {0};
        for (auto& s : SubLat) {
            double part[12];
            check(qcd_measure_fmunu_sums(s.ctx, get_F0mu ? 1 : 0, Fmunu_index1, Fmunu_index2, part), "qcd_measure_fmunu_sums");
            for (int i = 0; i < 12; i++) f[i] += part[i];
        }
        const double v = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
        for (int i = 0; i < 12; i++) fmunu_[i][index] = f[i] / v;
    }
}

// S(x), S(y), S(z) and the Polyakov loop plane by plane (lattice_analysis_Action_diff,
// lattice_analysis_PL_diff, SU(N)/suncl/suncl.cpp:1116-1260).  A plane of constant x lies in one part;
// planes of constant y or z are cut by the parts, whose sums are added before normalising with the
// full plane (for parts of equal width this is the reference's average over parts).
void model::measure_profiles(int index)
{
    const double g = lattice_group;
    for (int axis = 0; axis < 3; axis++) {
        const int ext = lattice_full_size[axis];
        std::vector<double> plq(2 * ext, 0.0), pl(2 * ext, 0.0), buf;
        int x0 = 0;
        for (auto& s : SubLat) {
            const int n = axis == 0 ? s.Nx : ext;
            const int at = axis == 0 ? x0 : 0;
            buf.assign(2 * n, 0.0);
            if (get_actions_diff) {
                check(qcd_measure_plaquette_profile_sums(s.ctx, axis, buf.data()), "qcd_measure_plaquette_profile_sums");
                for (int i = 0; i < 2 * n; i++) plq[2 * at + i] += buf[i];
            }
            if (PL_level > 2) {
                check(qcd_measure_polyakov_profile_sums(s.ctx, axis, buf.data()), "qcd_measure_polyakov_profile_sums");
                for (int i = 0; i < 2 * n; i++) pl[2 * at + i] += buf[i];
            }
            x0 += s.Nx;
        }
        const double v = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3] / ext;
        const double n2 = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] / ext;
        for (int b = 0; b < ext; b++) {
            if (get_actions_diff) {
                const double ss = BETA * (1 - plq[2 * b] / (3 * v) / g), st = BETA * (1 - plq[2 * b + 1] / (3 * v) / g);
                s_diff_[axis][(size_t)b * ITER + index] = 0.5 * (ss + st);   // S_total_diff
            }
            if (PL_level > 2) {
                pl_diff_[axis][(size_t)b * ITER + index] = pl[2 * b] / (n2 * g);
                pl_diff_im_[axis][(size_t)b * ITER + index] = pl[2 * b + 1] / (n2 * g);
            }
        }
    }
}

// the two "Differentiated ..." sections of the result file (SU(N)/suncl/suncl.cpp:1696-1790)
void model::write_profiles(FILE* stream)
{
    static const char* axis_name[3] = {"X", "Y", "Z"};
    auto row = [&](const char* label, const std::vector<double>& series, int ext, bool variance) {
        fprintf(stream, "\n%s", label);
        for (int b = 0; b < ext; b++) {
            data_analysis d;
            d.data.assign(series.begin() + (size_t)b * ITER, series.begin() + (size_t)(b + 1) * ITER);
            lattice_data_analysis(&d);
            fprintf(stream, "% 16.13e ", variance ? d.variance : d.mean_value);
        }
    };
    if (PL_level > 2) {
        fprintf(stream, " ***************************************************\n");
        fprintf(stream, "Differentiated Polyakov loop data (#, PL, PL_im, PL_variance, PL_im_variance):");
        fprintf(stream, "\n");
        for (int axis = 0; axis < 3; axis++) {
            const int ext = lattice_full_size[axis];
            fprintf(stream, "\n%s              ", axis_name[axis]);
            for (int i = 0; i < ext; i++) fprintf(stream, "%2u                   ", i);
            row("PL             ", pl_diff_[axis], ext, false);
            row("PL_im          ", pl_diff_im_[axis], ext, false);
            row("PL_variance    ", pl_diff_[axis], ext, true);
            row("PL_im_variance ", pl_diff_im_[axis], ext, true);
            fprintf(stream, "\n");
        }
    }
    if (get_actions_diff) {
        fprintf(stream, " ***************************************************\n");
        fprintf(stream, "Differentiated S data (#, S_total, S_variance):");
        fprintf(stream, "\n");
        for (int axis = 0; axis < 3; axis++) {
            const int ext = lattice_full_size[axis];
            fprintf(stream, "\n%s              ", axis_name[axis]);
            for (int i = 0; i < ext; i++) fprintf(stream, "%2u                   ", i);
            row("S             ", s_diff_[axis], ext, false);
            row("S_variance    ", s_diff_[axis], ext, true);
            fprintf(stream, "\n");
        }
    }
}

// the sample's table entries from its raw lattice sums
void model::derive_sample(int index)
{
    const double plq[2] = {plq_raw_[0][index], plq_raw_[1][index]};
    const double v = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
    const double n3 = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2];
    const double g = lattice_group;
    // SU(N)/suncpp/Measurements/Plq.h:90-92, S.h:92-97
    const double p3[3] = {plq[0] / (3 * v), plq[1] / (3 * v), (plq[0] + plq[1]) / (6 * v)};
    for (int i = 0; i < 3; i++) {
        plq_[i][index] = p3[i];
        act_[i][index] = BETA * (1 - p3[i] / g);
    }
    // SU(N)/suncl/su3cpu.cpp:1427-1430
    pl_[0][index] = pl_raw_[0][index] / (n3 * g);
    pl_[1][index] = pl_raw_[1][index] / (n3 * g);
    pl_[2][index] = pl_raw_[2][index] / (n3 * g * g);
    pl_[3][index] = pl_raw_[3][index] / (n3 * g * g * g * g);
}

// ---------------------------------------------------------------------------------------------
// lattice state files (SU(N)/suncl/suncl.cpp:1857-2079).  Layout of the reference's .qcg:
//   64-dword header | lattice_measurement (double2 x size_F) | lattice_energies (double2 x
//   energies_size: raw S sums per sample) | lattice_energies_plq (same, raw plaquette sums) |
//   [lattice_polyakov_loop (double2 x polyakov_loop_size: raw Re, Im sums)] | lattice_table
//   (double4 rows, reference device layout).
// After that this engine appends what the reference does not need (it replays its generator):
//   "QCDRLX1\0" | u64 streams | u64 draws | ctl | slot 0 | slot 1 of the per-thread streams in
//   GLOBAL stream order, so a file written by a run cut into parts loads into any other cut.

static const unsigned int BIN_HEADER_SIZE = 64;     // suncl.cpp:60
static const int MAX_SAMPLES = 1 << 22;             // ITER beyond this is refused (every sample series is held in memory)
static const unsigned int SIZE_ALIGN = 256;         // buffer_size_align step (clinterface.cpp:1580-1590): workgroup size
static unsigned int size_align(unsigned int n) { return (1 + (n - 1) / SIZE_ALIGN) * SIZE_ALIGN; }
static unsigned int convert_str_uint(const char* str, unsigned int offset)   // suncl.cpp:709-716
{
    unsigned int result = 0, pointer = offset;
    for (unsigned int i = 0; i < 4; i++)
        if (pointer < strlen(str)) result += ((unsigned int)(unsigned char)str[pointer++] << (i * 8));
    return result;
}
static void split_double(double x, unsigned int& low, unsigned int& high)
{
    unsigned long long u;
    memcpy(&u, &x, sizeof(u));
    low = (unsigned int)(u & 0xFFFFFFFFu);
    high = (unsigned int)(u >> 32);
}
static double join_double(unsigned int low, unsigned int high)
{
    const unsigned long long u = ((unsigned long long)high << 32) | low;
    double x;
    memcpy(&x, &u, sizeof(x));
    return x;
}

unsigned int model::energies_size() const { return size_align((unsigned int)ITER); }
// lattice_energies_plq holds the plaquette sums and, with F_mu_nu, six more blocks: index1 and
// index2, three planes each (MODEL_energies_size = 7, suncl.h:222; offsets suncl.cpp:1626-1627)
unsigned int model::energies_size_F() const { return (get_Fmunu || get_F0mu) ? 7u * energies_size() : energies_size(); }
unsigned int model::table_row_size() const
{
    return size_align((unsigned int)(lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3]));
}

std::vector<unsigned int> model::lattice_make_bin_header(void) const   // suncl.cpp:1939-1988
{
    std::vector<unsigned int> r(BIN_HEADER_SIZE, 0u);
    const char* bin_prefix = "QCDGPU";
    int k = 0;
    for (int i = 0; i < (int)ceil((double)strlen(bin_prefix) / 4); i++) r[k++] = convert_str_uint(bin_prefix, i * 4);
    k = 2;
    for (int i = 0; i < (int)ceil((double)version.size() / 4); i++) r[k++] = convert_str_uint(version.c_str(), i * 4);
    k = 4;
    r[k++] = (unsigned int)INIT;
    r[k++] = ints == model_start_hot ? 0u : (ints == model_start_cold ? 1u : 2u);
    r[k++] = PRNG0->PRNG_randseries;
    r[k++] = 2u + (unsigned int)PRNG0->PRNG_generator;   // RANLUX0..4 = 2..6 (random.cpp:110-120)
    static const unsigned int nskip[5] = {24, 48, 97, 223, 389};
    r[k++] = nskip[PRNG0->lux_level()];
    r[k++] = PRNG_counter;
    r[k++] = (unsigned int)NAV;
    r[k++] = (unsigned int)NAV_counter;
    r[k++] = (unsigned int)NITER;
    r[k++] = (unsigned int)ITER;
    r[k++] = (unsigned int)ITER_counter;
    r[k++] = (unsigned int)NHIT;
    r[k++] = (unsigned int)wilson_R;
    r[k++] = (unsigned int)wilson_T;
    r[k++] = precision_word();   // model::convert_precision_to_uint, SU(N)/suncl/suncl.cpp:732-737
    split_double(BETA, r[k], r[k + 1]); k += 2;
    split_double(PHI, r[k], r[k + 1]); k += 2;
    split_double(OMEGA, r[k], r[k + 1]); k += 2;
    r[k++] = (unsigned int)PL_level;
    r[k++] = (unsigned int)Fmunu_index1;
    r[k++] = (unsigned int)Fmunu_index2;
    r[k++] = get_Fmunu ? 1u : 0u;
    r[k++] = get_actions_avr ? 1u : 0u;
    r[k++] = get_plaquettes_avr ? 1u : 0u;
    r[k++] = get_wilson_loop ? 1u : 0u;
    r[k++] = 0u;   // lattice_measurement_size_F: the reduce scratch buffer is not part of this engine's state
    r[k++] = energies_size();
    r[k++] = energies_size_F();   // lattice_energies_size_F: seven blocks when F_mu_nu is measured (suncl.cpp:2283-2285)
    r[k++] = energies_size();   // lattice_polyakov_loop_size
    r[k++] = (unsigned int)lattice_group;
    r[k++] = (unsigned int)lattice_nd;
    for (int i = 0; i < lattice_nd; i++) r[k++] = (unsigned int)lattice_full_size[i];
    for (int i = 0; i < lattice_nd; i++) r[k++] = (unsigned int)lattice_domain_size[i];
    // the reference leaves the rest of the header zero and does not read it; the last word says how many sites share
    // a random stream in this engine (the generator states behind the link table are laid out by it)
    r[BIN_HEADER_SIZE - 1] = (unsigned int)rng_share_used_;
    return r;
}

bool model::lattice_load_bin_header(const unsigned int* head)   // suncl.cpp:1990-2041
{
    const char* bin_prefix = "QCDGPU";
    bool ok = true;
    int k = 0;
    for (int i = 0; i < (int)ceil((double)strlen(bin_prefix) / 4); i++) ok &= (head[k++] == convert_str_uint(bin_prefix, i * 4));
    k = 2;
    for (int i = 0; i < (int)ceil((double)version.size() / 4); i++) ok &= (head[k++] == convert_str_uint(version.c_str(), i * 4));
    if (!ok) return false;
    k = 5;
    ints = head[k] == 0 ? model_start_hot : (head[k] == 2 ? model_start_gid : model_start_cold); k++;
    PRNG0->PRNG_randseries = head[k++];
    const unsigned int gen = head[k++];
    if (gen < 2 || gen > 6) return false;
    PRNG0->PRNG_generator = (PRNG::PRNG_generators)(gen - 2);
    k++;   // RL_nskip follows from the generator
    PRNG_counter = head[k++];
    NAV = (int)head[k++];
    NAV_counter = (int)head[k++];
    NITER = (int)head[k++];
    ITER = (int)head[k++];
    ITER_counter = (int)head[k++];
    NHIT = (int)head[k++];
    wilson_R = (int)head[k++];
    wilson_T = (int)head[k++];
    if (head[k++] != precision_word()) return false;   // a state of another precision: its link table has other element sizes
    BETA = join_double(head[k], head[k + 1]); k += 2;
    PHI = join_double(head[k], head[k + 1]); k += 2;
    OMEGA = join_double(head[k], head[k + 1]); k += 2;
    PL_level = (int)head[k++];
    Fmunu_index1 = (int)head[k++];
    Fmunu_index2 = (int)head[k++];
    get_Fmunu = head[k++] != 0;          // (F0MU is not in the header: the run's init file says so, as for the reference)
    get_actions_avr = head[k++] != 0;
    get_plaquettes_avr = head[k++] != 0;
    get_wilson_loop = head[k++] != 0;
    state_sizes_[0] = head[k++];
    state_sizes_[1] = head[k++];
    state_sizes_[2] = head[k++];
    state_sizes_[3] = head[k++];
    lattice_group = (int)head[k++];
    lattice_nd = (int)head[k++];
    if (lattice_nd != 4) return false;
    for (int i = 0; i < lattice_nd; i++) lattice_full_size[i] = (int)head[k++];
    for (int i = 0; i < lattice_nd; i++) lattice_domain_size[i] = (int)head[k++];
    // (zero in a file the reference wrote: one stream per site pair, and no generator states to lay out)
    const unsigned int share = head[BIN_HEADER_SIZE - 1];
    if (share == 1 || share == 2 || share == 4) RNGSHARE = (int)share;
    // A header is trusted with the size of everything that is allocated next: words that no run of this program or of
    // the reference can have written make it a bad header, not a request for memory
    if (ITER < 1 || ITER > MAX_SAMPLES || ITER_counter < 0 || ITER_counter > ITER || NAV < 0 || NAV_counter < 0 || NAV_counter > NAV || NITER < 0 ||
        NHIT < 1 || NHIT > 4096 || wilson_R < 0 || wilson_T < 0 || wilson_R > 4096 || wilson_T > 4096)
        return false;
    if (lattice_group != 2 && lattice_group != 3) return false;
    unsigned long long sites = 1;
    for (int i = 0; i < 4; i++) {
        if (lattice_full_size[i] < 2 || lattice_full_size[i] > 4096 || lattice_domain_size[i] != lattice_full_size[i]) return false;
        sites *= (unsigned long long)lattice_full_size[i];
    }
    if (sites > (1ull << 30)) return false;
    if (state_sizes_[1] != energies_size() || (state_sizes_[2] != state_sizes_[1] && state_sizes_[2] != 7u * state_sizes_[1]) ||
        state_sizes_[3] != state_sizes_[1] || state_sizes_[0] > 7ull * (sites + 4096))
        return false;
    return true;
}

void model::lattice_save_state(void)
{
    time_t now;
    char* timesave = current_datetime(&now);
    char name[64] = "";
    snprintf(name, sizeof(name), "%.2s-%.3s-%.2s-%.2s-%.2s-%.2s.qcg", timesave + 22, timesave + 4, timesave + 8,
             timesave + 11, timesave + 14, timesave + 17);
    free(timesave);
    state_file = path + fprefix + name;
    FILE* stream = fopen(state_file.c_str(), "wb");
    if (!stream) throw model_error("cannot write " + state_file);
    const std::vector<unsigned int> head = lattice_make_bin_header();
    fwrite(head.data(), sizeof(unsigned int), BIN_HEADER_SIZE, stream);

    const double v = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
    const unsigned int es = energies_size();
    std::vector<double> sec(2 * (size_t)es);
    auto write_pairs = [&](const std::vector<double>& a, const std::vector<double>& b, double factor) {
        std::fill(sec.begin(), sec.end(), 0.0);
        for (int i = 0; i < ITER_counter && i < ITER; i++) {
            sec[2 * i] = a[i] * factor;
            sec[2 * i + 1] = b[i] * factor;
        }
        fwrite(sec.data(), 2 * sizeof(double), es, stream);
    };
    write_pairs(act_[0], act_[1], 3.0 * v);                               // lattice_energies: S x its denominator (suncl.cpp:1340-1348)
    if (get_plaquettes_avr || get_Fmunu || get_F0mu) {                    // lattice_energies_plq (suncl.cpp:1892): lattice sums
        write_pairs(plq_raw_[0], plq_raw_[1], 1.0);
        if (get_Fmunu || get_F0mu)                                        // blocks 1-3: index1, blocks 4-6: index2, per plane (re, im)
            for (int a = 0; a < 2; a++)
                for (int pl = 0; pl < 3; pl++) write_pairs(fmunu_[6 * a + 2 * pl], fmunu_[6 * a + 2 * pl + 1], v);
    }
    if (get_wilson_loop) {                                                // lattice_wilson_loop: doubles, sums over 3V
        std::vector<double> w((size_t)es, 0.0);
        for (int i = 0; i < ITER_counter && i < ITER; i++) w[i] = wl_[i] * 3.0 * v;
        fwrite(w.data(), sizeof(double), es, stream);
    }
    if (PL_level > 0) write_pairs(pl_raw_[0], pl_raw_[1], 1.0);           // lattice_polyakov_loop: Re, Im sums

    // link table: rows of the whole lattice, each part's sites at their global position
    const int rows = (lattice_group == 3) ? 12 : 4;
    const unsigned int rs = table_row_size();
    const size_t gslice = (size_t)lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
    const size_t qb = 4 * real_bytes();   // bytes of a quad: double4 or float4 rows (SU(N)/suncl/model.cl:140-162)
    std::vector<unsigned char> table((size_t)rows * rs * qb, 0);
    size_t x0 = 0;
    for (auto& s : SubLat) {
        const size_t nloc = (size_t)s.Nx * gslice;
        std::vector<unsigned char> part((size_t)rows * nloc * qb);
        check(qcd_download(s.ctx, part.data(), part.size(), (int)nloc), "qcd_download");
        for (int r = 0; r < rows; r++)
            memcpy(&table[((size_t)r * rs + x0 * gslice) * qb], &part[(size_t)r * nloc * qb], nloc * qb);
        x0 += (size_t)s.Nx;
    }
    fwrite(table.data(), qb, (size_t)rows * rs, stream);

    // per-thread generator state in global stream order (rng_share_used_ sites of one parity per stream)
    const size_t NS = (size_t)lattice_full_size[0] * gslice / 2 / (size_t)rng_share_used_;
    std::vector<unsigned int> ctl(NS), slots(2 * 25 * NS);
    unsigned long long draws = 0;
    size_t first = 0;
    for (auto& s : SubLat) {
        const size_t ns = (size_t)s.Nx * gslice / 2 / (size_t)rng_share_used_;
        std::vector<unsigned char> st(qcd_rng_state_bytes(s.ctx));
        check(qcd_rng_state_download(s.ctx, st.data(), st.size()), "qcd_rng_state_download");
        unsigned long long d;
        memcpy(&d, st.data(), sizeof(d));
        draws += d;
        const unsigned int* w = (const unsigned int*)(st.data() + sizeof(d));
        memcpy(&ctl[first], w, ns * sizeof(unsigned int));
        w += ns;
        for (int slot = 0; slot < 2; slot++)
            for (int k = 0; k < 25; k++, w += ns)
                memcpy(&slots[((size_t)slot * 25 + k) * NS + first], w, ns * sizeof(unsigned int));
        first += ns;
    }
    const char magic[8] = {'Q', 'C', 'D', 'R', 'L', 'X', '1', 0};
    const unsigned long long ns64 = NS;
    fwrite(magic, 1, 8, stream);
    fwrite(&ns64, sizeof(ns64), 1, stream);
    fwrite(&draws, sizeof(draws), 1, stream);
    fwrite(ctl.data(), sizeof(unsigned int), ctl.size(), stream);
    fwrite(slots.data(), sizeof(unsigned int), slots.size(), stream);
    // the reference's Polyakov section holds Re and Im only; |P|^2 and |P|^4 sums follow here
    const char magic2[8] = {'Q', 'C', 'D', 'P', 'L', '2', '4', 0};
    const unsigned long long n64 = (unsigned long long)ITER;
    fwrite(magic2, 1, 8, stream);
    fwrite(&n64, sizeof(n64), 1, stream);
    fwrite(pl_raw_[2].data(), sizeof(double), (size_t)ITER, stream);
    fwrite(pl_raw_[3].data(), sizeof(double), (size_t)ITER, stream);
    // the plane-resolved series: per axis the lengths and the values of S_total_diff, PL_diff, PL_im_diff
    const char magic3[8] = {'Q', 'C', 'D', 'P', 'R', 'F', '1', 0};
    fwrite(magic3, 1, 8, stream);
    fwrite(&n64, sizeof(n64), 1, stream);
    for (int a = 0; a < 3; a++)
        for (const std::vector<double>* v : {&s_diff_[a], &pl_diff_[a], &pl_diff_im_[a]}) {
            const unsigned long long len = v->size();
            fwrite(&len, sizeof(len), 1, stream);
            if (len) fwrite(v->data(), sizeof(double), v->size(), stream);   // (an empty series has no storage to point at)
        }
    if (fclose(stream)) printf("The file was not closed!\n");
    time(&last_save_);
}

// Two calls, as in the reference: the first (from lattice_init, before the contexts exist) reads
// the header and sets the run parameters; the second (from lattice_simulate) reads the samples,
// the configuration and the generator state into the contexts.
void model::lattice_load_state(void)
{
    const std::string file = path + fstate;
    FILE* stream = fopen(file.c_str(), "rb");
    if (!stream) throw model_error("cannot read state file " + file);
    std::vector<unsigned int> head(BIN_HEADER_SIZE);
    bool ok = fread(head.data(), sizeof(unsigned int), BIN_HEADER_SIZE, stream) == BIN_HEADER_SIZE;
    if (LOAD_state == 0) {
        ok = ok && lattice_load_bin_header(head.data());
        fclose(stream);
        if (!ok) throw model_error("[ERROR in header!!!] " + file);
        LOAD_state++;
        return;
    }
    const double v = (double)lattice_full_size[0] * lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
    const double g = lattice_group;
    {
        // the sections this run expects, and the link table behind them, must fit the file: a file written
        // with other observable switches has its link table at another offset
        const bool plq_sec = get_plaquettes_avr || get_Fmunu || get_F0mu;
        unsigned long long need = BIN_HEADER_SIZE * sizeof(unsigned int) + 16ull * state_sizes_[0] + 16ull * state_sizes_[1];
        if (plq_sec) need += 16ull * state_sizes_[2];
        if (get_wilson_loop) need += 8ull * state_sizes_[1];
        if (PL_level > 0) need += 16ull * state_sizes_[3];
        need += 4ull * real_bytes() * ((lattice_group == 3) ? 12 : 4) * table_row_size();
        const long at = ftell(stream);
        fseek(stream, 0, SEEK_END);
        const unsigned long long have = (unsigned long long)ftell(stream);
        fseek(stream, at, SEEK_SET);
        if (have < need) {
            fclose(stream);
            throw model_error("state file " + file + " is shorter than its header and this run's observable switches imply");
        }
    }
    std::vector<double> sec;
    auto read_pairs = [&](unsigned int count, std::vector<double>* a, std::vector<double>* b) {
        sec.assign(2 * (size_t)count, 0.0);
        ok = ok && fread(sec.data(), 2 * sizeof(double), count, stream) == count;
        for (int i = 0; a && i < ITER_counter && i < ITER && (unsigned int)i < count; i++) {
            (*a)[i] = sec[2 * i];
            (*b)[i] = sec[2 * i + 1];
        }
    };
    std::vector<double> s_raw[2] = {std::vector<double>(ITER, 0.0), std::vector<double>(ITER, 0.0)};
    read_pairs(state_sizes_[0], nullptr, nullptr);                                      // lattice_measurement
    read_pairs(state_sizes_[1], &s_raw[0], &s_raw[1]);                                  // lattice_energies
    const bool plq_section = get_plaquettes_avr || get_Fmunu || get_F0mu;               // suncl.cpp:2063
    if (plq_section) {                                                                  // lattice_energies_plq
        const bool with_F = (get_Fmunu || get_F0mu) && state_sizes_[2] == 7u * state_sizes_[1];
        const unsigned int blk = with_F ? state_sizes_[1] : state_sizes_[2];
        read_pairs(blk, &plq_raw_[0], &plq_raw_[1]);
        if (with_F)
            for (int a = 0; a < 2; a++)
                for (int pl = 0; pl < 3; pl++) {
                    std::vector<double> re(ITER, 0.0), im(ITER, 0.0);
                    read_pairs(blk, &re, &im);
                    for (int i = 0; i < ITER_counter && i < ITER && !fmunu_[0].empty(); i++) {
                        fmunu_[6 * a + 2 * pl][i] = re[i] / v;
                        fmunu_[6 * a + 2 * pl + 1][i] = im[i] / v;
                    }
                }
    }
    if (!plq_section)
        for (int i = 0; i < ITER_counter && i < ITER; i++)   // plaquette sums back from S = beta (1 - Plq / N)
            for (int c = 0; c < 2; c++) plq_raw_[c][i] = 3.0 * v * g * (1.0 - s_raw[c][i] / (3.0 * v) / BETA);
    if (get_wilson_loop) {                                                              // lattice_wilson_loop
        std::vector<double> w((size_t)state_sizes_[1], 0.0);
        ok = ok && fread(w.data(), sizeof(double), w.size(), stream) == w.size();
        for (int i = 0; i < ITER_counter && i < ITER && (size_t)i < w.size(); i++) wl_[i] = w[i] / (3.0 * v);
    }
    if (PL_level > 0) read_pairs(state_sizes_[3], &pl_raw_[0], &pl_raw_[1]);            // lattice_polyakov_loop
    const int rows = (lattice_group == 3) ? 12 : 4;
    const unsigned int rs = table_row_size();
    const size_t gslice = (size_t)lattice_full_size[1] * lattice_full_size[2] * lattice_full_size[3];
    const size_t qb = 4 * real_bytes();
    std::vector<unsigned char> table((size_t)rows * rs * qb);
    ok = ok && fread(table.data(), qb, (size_t)rows * rs, stream) == (size_t)rows * rs;
    if (!ok) {
        fclose(stream);
        throw model_error("state file is truncated: " + file);
    }
    size_t x0 = 0;
    for (auto& s : SubLat) {
        const size_t nloc = (size_t)s.Nx * gslice;
        std::vector<unsigned char> part((size_t)rows * nloc * qb);
        for (int r = 0; r < rows; r++)
            memcpy(&part[(size_t)r * nloc * qb], &table[((size_t)r * rs + x0 * gslice) * qb], nloc * qb);
        check(qcd_init_lattice(s.ctx), "qcd_init_lattice");   // tables of the start mode (gid), stream seeds
        check(qcd_upload(s.ctx, part.data(), part.size(), (int)nloc), "qcd_upload");
        x0 += (size_t)s.Nx;
    }
    // generator state (absent in files the reference wrote: the streams then start from their seeds)
    char magic[8];
    unsigned long long ns64 = 0, draws = 0;
    const size_t NS = (size_t)lattice_full_size[0] * gslice / 2 / (size_t)rng_share_used_;
    if (fread(magic, 1, 8, stream) == 8 && !memcmp(magic, "QCDRLX1", 8) && fread(&ns64, sizeof(ns64), 1, stream) == 1 &&
        fread(&draws, sizeof(draws), 1, stream) == 1 && ns64 == NS) {
        std::vector<unsigned int> ctl(NS), slots(2 * 25 * NS);
        if (fread(ctl.data(), sizeof(unsigned int), NS, stream) != NS ||
            fread(slots.data(), sizeof(unsigned int), slots.size(), stream) != slots.size()) {
            fclose(stream);
            throw model_error("generator state is truncated: " + file);
        }
        size_t first = 0;
        bool first_part = true;
        for (auto& s : SubLat) {
            const size_t ns = (size_t)s.Nx * gslice / 2 / (size_t)rng_share_used_;
            std::vector<unsigned char> st(qcd_rng_state_bytes(s.ctx));
            const unsigned long long d = first_part ? draws : 0ull;   // the total is kept on the first part
            memcpy(st.data(), &d, sizeof(d));
            unsigned int* w = (unsigned int*)(st.data() + sizeof(d));
            memcpy(w, &ctl[first], ns * sizeof(unsigned int));
            w += ns;
            for (int slot = 0; slot < 2; slot++)
                for (int k = 0; k < 25; k++, w += ns)
                    memcpy(w, &slots[((size_t)slot * 25 + k) * NS + first], ns * sizeof(unsigned int));
            check(qcd_rng_state_upload(s.ctx, st.data(), st.size()), "qcd_rng_state_upload");
            first += ns;
            first_part = false;
        }
    } else {
        printf("state file carries no generator state: streams start from their seeds\n");
    }
    char magic2[8];
    unsigned long long n64 = 0;
    if (fread(magic2, 1, 8, stream) == 8 && !memcmp(magic2, "QCDPL24", 8) && fread(&n64, sizeof(n64), 1, stream) == 1 &&
        n64 == (unsigned long long)ITER) {
        if (fread(pl_raw_[2].data(), sizeof(double), (size_t)ITER, stream) != (size_t)ITER ||
            fread(pl_raw_[3].data(), sizeof(double), (size_t)ITER, stream) != (size_t)ITER)
            printf("state file: |P|^2, |P|^4 sums are truncated\n");
    }
    bool profiles_read = false;
    if (fread(magic2, 1, 8, stream) == 8 && !memcmp(magic2, "QCDPRF1", 8) && fread(&n64, sizeof(n64), 1, stream) == 1 &&
        n64 == (unsigned long long)ITER) {
        profiles_read = true;
        for (int a = 0; a < 3 && profiles_read; a++)
            for (std::vector<double>* v : {&s_diff_[a], &pl_diff_[a], &pl_diff_im_[a]}) {
                unsigned long long len = 0;
                if (fread(&len, sizeof(len), 1, stream) != 1) { profiles_read = false; break; }
                if (len > 64ull * (unsigned long long)ITER) { profiles_read = false; break; }   // not a length this program writes
                std::vector<double> tmp((size_t)len);
                if (len && fread(tmp.data(), sizeof(double), tmp.size(), stream) != tmp.size()) { profiles_read = false; break; }
                if (v->empty()) continue;                       // series not asked for in this run
                if (tmp.size() != v->size()) { profiles_read = false; break; }   // asked for, but not in the file
                *v = tmp;
            }
    }
    fclose(stream);
    if (!profiles_read && (get_actions_diff || PL_level > 2))
        throw model_error("state file does not carry the plane-resolved series this run asks for (GETSDIFF / PL_LEVEL = 3)");
    for (int i = 0; i < ITER_counter && i < ITER; i++) derive_sample(i);
    // halo slices of the parts
    for (int dir = 0; dir < 4; dir++)
        for (int parity = 0; parity < 2; parity++) UpdateEdges(dir, parity);
    LOAD_state++;
}

// lattice_simulate (SU(N)/suncl/suncl.cpp:5212-5505): sample 0 on the initial configuration,
// NAV thermalisation sweeps, then ITER-1 times { NITER sweeps ; measure }
void model::lattice_simulate(void)
{
    if (SubLat.empty() || !SubLat[0].ctx) throw model_error("lattice_simulate before lattice_init");
    if (ITER < 1) throw model_error("ITER < 1");
    if (ITER > MAX_SAMPLES) throw model_error("ITER beyond 2^22 samples: every sample series is held in memory");
    if (NAV < 0 || NITER < 0) throw model_error("NAV or NITER negative");
    for (int i = 0; i < 3; i++) {
        plq_[i].assign(ITER, 0.0);
        act_[i].assign(ITER, 0.0);
    }
    for (int i = 0; i < 4; i++) pl_[i].assign(ITER, 0.0);
    for (int i = 0; i < 2; i++) plq_raw_[i].assign(ITER, 0.0);
    for (int i = 0; i < 4; i++) pl_raw_[i].assign(ITER, 0.0);
    wl_.assign(ITER, 0.0);
    for (int i = 0; i < 12; i++) fmunu_[i].assign((get_Fmunu || get_F0mu) ? ITER : 0, 0.0);
    for (int a = 0; a < 3; a++) {
        s_diff_[a].assign(get_actions_diff ? (size_t)lattice_full_size[a] * ITER : 0, 0.0);
        pl_diff_[a].assign(PL_level > 2 ? (size_t)lattice_full_size[a] * ITER : 0, 0.0);
        pl_diff_im_[a].assign(PL_level > 2 ? (size_t)lattice_full_size[a] * ITER : 0, 0.0);
    }
    free(timestart);
    timestart = current_datetime(&ltimestart);
    time(&last_save_);
    if (INIT != 0) {
        for (auto& s : SubLat) check(qcd_init_lattice(s.ctx), "qcd_init_lattice");
        // hot start: a part's halo slices hold its neighbours' random links only after an exchange
        if (ints == model_start_hot)
            for (int dir = 0; dir < 4; dir++)
                for (int parity = 0; parity < 2; parity++) UpdateEdges(dir, parity);
        NAV_counter = 0;
        ITER_counter = 0;
        PRNG_counter = 0;
        measure(ITER_counter++);
        if (!turnoff_config_save) lattice_save_state();
    } else {
        lattice_load_state();   // samples so far, configuration, generator state (suncl.cpp:3956, 5349-5354)
        printf("NAV_start=%i, ITER_start=%i\n", NAV_counter, ITER_counter);
    }
    // thermalisation (suncl.cpp:5356-5388)
    for (int i = NAV_counter; i < NAV; i++) {
        sweep();
        if (i % 10 == 0) printf("\rGPU thermalization [%i]", i);
        NAV_counter++;
        if (!turnoff_config_save && difftime(time(NULL), last_save_) > write_lattice_state_every_secs) {
            lattice_save_state();
            printf("\ncurrent configuration saved\n");
        }
    }
    // working cycles (suncl.cpp:5390-5495)
    for (int i = ITER_counter; i < ITER; i++) {
        for (int j = 0; j < NITER; j++) sweep(true);
        if (i % 10 == 0) printf("\rGPU working iteration [%u]", i);
        measure(ITER_counter++);
        if (!turnoff_config_save && difftime(time(NULL), last_save_) > write_lattice_state_every_secs) {
            lattice_save_state();
            printf("\ncurrent configuration saved\n");
        }
        if (stop_at_sample > 0 && ITER_counter >= stop_at_sample && ITER_counter < ITER) {
            stopped_early = true;
            break;
        }
    }
    if (!turnoff_config_save) lattice_save_state();
    for (auto& s : SubLat) check(qcd_synchronize(s.ctx), "qcd_synchronize");
    free(timeend);
    timeend = current_datetime(&ltimeend);
    printf("\rGPU simulations are done\n");
}

void model::lattice_analysis(void)
{
    Analysis[DM_Plq_spat].data = plq_[0];
    Analysis[DM_Plq_temp].data = plq_[1];
    Analysis[DM_Plq_total].data = plq_[2];
    Analysis[DM_S_spat].data = act_[0];
    Analysis[DM_S_temp].data = act_[1];
    Analysis[DM_S_total].data = act_[2];
    Analysis[DM_Polyakov_loop].data = pl_[0];
    Analysis[DM_Polyakov_loop_im].data = pl_[1];
    Analysis[DM_Polyakov_loop_P2].data = pl_[2];
    Analysis[DM_Polyakov_loop_P4].data = pl_[3];
    Analysis[DM_Wilson_loop].data = wl_;
    lattice_data_analysis(&Analysis[DM_Wilson_loop]);
    if (get_Fmunu || get_F0mu) {
        // names and the |F| columns: SU(N)/suncl/suncl.cpp:1004-1063, data_analysis.cpp:186-206
        static const char* xyz[3] = {"xy", "xz", "yz"};
        static const char* xyzt[3] = {"xt", "yt", "zt"};
        static const char* reim[2] = {"re", "im"};
        for (int i = 0; i < 6; i++) {
            char buf[32];
            snprintf(buf, sizeof(buf), "Fmunu_%s_%u_%s", get_Fmunu ? xyz[i / 2] : xyzt[i / 2], Fmunu_index1, reim[i & 1]);
            fmunu_names_[i] = buf;
            snprintf(buf, sizeof(buf), "Fmunu_%s_%u_%s", get_Fmunu ? xyz[i / 2] : xyzt[i / 2], Fmunu_index2, reim[i & 1]);
            fmunu_names_[6 + i] = buf;
            Analysis[DM_Fmunu_3 + i].data = fmunu_[i];
            Analysis[DM_Fmunu_8 + i].data = fmunu_[6 + i];
        }
        for (int a = 0; a < 2; a++)
            for (int c = 0; c < 2; c++) {
                char buf[32];
                snprintf(buf, sizeof(buf), "%s_abs_%1u_%s", get_Fmunu ? "Fmunu" : "F0mu", a ? Fmunu_index2 : Fmunu_index1, reim[c]);
                fmunu_names_[12 + 2 * a + c] = buf;
                std::vector<double>& d = Analysis[DM_Fmunu_abs_3_re + 2 * a + c].data;
                d.assign(ITER, 0.0);
                for (int i = 0; i < ITER; i++) {
                    const double x = fmunu_[6 * a + c][i], y = fmunu_[6 * a + 2 + c][i], z = fmunu_[6 * a + 4 + c][i];
                    d[i] = sqrt(x * x + y * y + z * z);
                }
            }
        for (int i = 0; i < 16; i++) Analysis[DM_Fmunu_3 + i].data_name = fmunu_names_[i].c_str();
    }
    for (int i = 1; i <= DM_max; i++)
        if (!Analysis[i].data.empty()) lattice_data_analysis(&Analysis[i]);
}

const char* model::lattice_make_header2(void)
{
    // SU(N)/suncl/suncl.cpp:797-865, restricted to the observables of this engine
    char buf[8192];
    int j = 0;
    const int n = (int)sizeof(buf);
    const double dif = difftime(ltimeend, ltimestart);
    const int elapsdays = (int)(dif / 24 / 3600);
    const int elapshours = (((int)dif / 3600) % 3600);
    const int elapsminites = (((int)dif / 60) % 60);
    const int elapsseconds = (((int)dif) % 60);
    j += snprintf(buf + j, n - j, " Start time               : %s", timestart ? timestart : "\n");
    j += snprintf(buf + j, n - j, " Finish time              : %s", timeend ? timeend : "\n");
    j += snprintf(buf + j, n - j, " Elapsed time             : %i:%2.2i:%2.2i:%2.2i\n", elapsdays, elapshours, elapsminites, elapsseconds);
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    static const int order[] = {DM_S_total, DM_Plq_total, DM_Polyakov_loop, DM_Polyakov_loop_im,
                                DM_Polyakov_loop_P2, DM_Polyakov_loop_P4};
    for (int idx : order) {
        j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[idx].data_name, Analysis[idx].mean_value);
        j += snprintf(buf + j, n - j, " Variance %-16s: % 16.13e\n", Analysis[idx].data_name, Analysis[idx].variance);
    }
    if (get_wilson_loop) {
        j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_Wilson_loop].data_name, Analysis[DM_Wilson_loop].mean_value);
        j += snprintf(buf + j, n - j, " Variance %-16s: % 16.13e\n", Analysis[DM_Wilson_loop].data_name, Analysis[DM_Wilson_loop].variance);
    }
    j += snprintf(buf + j, n - j, " ***************************************************\n");
    if (get_Fmunu || get_F0mu) {
        for (int i = 0; i < 16; i++)
            j += snprintf(buf + j, n - j, " Mean %-20s: % 16.13e\n", Analysis[DM_Fmunu_3 + i].data_name, Analysis[DM_Fmunu_3 + i].mean_value);
        for (int i = 0; i < 16; i++)
            j += snprintf(buf + j, n - j, " Variance %-16s: % 16.13e\n", Analysis[DM_Fmunu_3 + i].data_name, Analysis[DM_Fmunu_3 + i].variance);
        j += snprintf(buf + j, n - j, " ***************************************************\n");
    }
    for (int idx : order)
        j += snprintf(buf + j, n - j, " GPU last %-16s: % 16.13e\n", Analysis[idx].data_name, Analysis[idx].GPU_last_value);
    header += buf;
    return header.c_str();
}

// result file (SU(N)/suncl/suncl.cpp:1679-1854): <path><prefix>YY-Mon-DD-HH-MM-SS.txt
void model::lattice_write_results(void)
{
    const char* header2 = lattice_make_header2();
    printf("%s\n", header2);
    char name[64] = "";
    if (timeend && strlen(timeend) >= 24)
        snprintf(name, sizeof(name), "%.2s-%.3s-%.2s-%.2s-%.2s-%.2s.txt", timeend + 22, timeend + 4, timeend + 8,
                 timeend + 11, timeend + 14, timeend + 17);
    result_file = path + fprefix + name;
    FILE* stream = fopen(result_file.c_str(), "w+");
    if (!stream) throw model_error("cannot write " + result_file);
    fputs(header.c_str(), stream);
    write_profiles(stream);
    fprintf(stream, " ***************************************************\n");
    fprintf(stream, " Data fields:\n");
    fprintf(stream, "    #,  ");
    std::vector<int> cols;
    if (get_plaquettes_avr) { cols.push_back(DM_Plq_spat); cols.push_back(DM_Plq_temp); cols.push_back(DM_Plq_total); }
    if (get_wilson_loop) cols.push_back(DM_Wilson_loop);   // column order of SU(N)/suncl/suncl.cpp:1796-1818
    if (get_actions_avr) { cols.push_back(DM_S_spat); cols.push_back(DM_S_temp); cols.push_back(DM_S_total); }
    if (PL_level > 0) { cols.push_back(DM_Polyakov_loop); cols.push_back(DM_Polyakov_loop_im); }
    if (PL_level > 1) { cols.push_back(DM_Polyakov_loop_P2); cols.push_back(DM_Polyakov_loop_P4); }
    if (get_Fmunu || get_F0mu)
        for (int i = 0; i < 16; i++) cols.push_back(DM_Fmunu_3 + i);   // suncl.cpp:1812-1818, 1845-1848
    for (int c : cols) fprintf(stream, "%-21s ", Analysis[c].data_name);
    fprintf(stream, "\n");
    fprintf(stream, " ***************************************************\n");
    for (int i = 0; i < ITER; i++) {
        fprintf(stream, "%5i", i);
        for (int c : cols) fprintf(stream, " % 16.13e", Analysis[c].data[i]);
        fprintf(stream, "\n");
    }
    if (fclose(stream)) printf("The file was not closed!\n");
}

// SU(N)/suncl/suncl.cpp:867-905: first five and the last sample
void model::lattice_print_measurements(void)
{
    printf("\n #        ");
    if (get_plaquettes_avr) printf(" %-11s %-11s", Analysis[DM_Plq_spat].data_name, Analysis[DM_Plq_temp].data_name);
    if (get_actions_avr) printf(" %-11s %-11s", Analysis[DM_S_spat].data_name, Analysis[DM_S_temp].data_name);
    if (PL_level > 0) printf(" %-11s %-11s", Analysis[DM_Polyakov_loop].data_name, Analysis[DM_Polyakov_loop_im].data_name);
    if (get_wilson_loop) printf(" %-11s", Analysis[DM_Wilson_loop].data_name);
    printf("\n");
    for (int i = 0; i < ITER; i++) {
        if ((i < 5) || (i == (ITER - 1))) {
            printf("[%5u]: ", i);
            if (get_plaquettes_avr) printf(" % 10.8f % 10.8f", Analysis[DM_Plq_spat].data[i], Analysis[DM_Plq_temp].data[i]);
            if (get_actions_avr) printf(" % 10.8f % 10.8f", Analysis[DM_S_spat].data[i], Analysis[DM_S_temp].data[i]);
            if (PL_level > 0) printf(" % 10.8f % 10.8f", Analysis[DM_Polyakov_loop].data[i], Analysis[DM_Polyakov_loop_im].data[i]);
            if (get_wilson_loop) printf(" % 10.8f", Analysis[DM_Wilson_loop].data[i]);
            printf("\n");
        }
    }
}

// per-kernel-class timing report (GPU::print_time_detailed, SU(N)/clinterface/clinterface.cpp:1626-1667)
void model::print_time_detailed(void)
{
    if (!profiling) return;
    printf("\n kernel class      launches        total ms      avg us\n");
    for (size_t k = 0; k < SubLat.size(); k++) {
        qcd_timing t;
        if (qcd_get_timing(SubLat[k].ctx, &t) != QCD_OK) continue;
        printf(" part %zu update   %10llu  %14.3f  %10.2f\n", k + 1, t.update_launches, t.update_ms,
               t.update_launches ? 1e3 * t.update_ms / t.update_launches : 0.0);
        printf(" part %zu measure  %10llu  %14.3f  %10.2f\n", k + 1, t.measure_launches, t.measure_ms,
               t.measure_launches ? 1e3 * t.measure_ms / t.measure_launches : 0.0);
        printf(" part %zu edges    %10llu  %14.3f  %10.2f\n", k + 1, t.halo_launches, t.halo_ms,
               t.halo_launches ? 1e3 * t.halo_ms / t.halo_launches : 0.0);
    }
}

}  // namespace model_HIP
