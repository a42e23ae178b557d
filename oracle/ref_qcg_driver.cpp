// ref_qcg_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Reads a lattice state file (.qcg) with the reference's OWN reader, model::lattice_load_state
// (SU(N)/suncl/suncl.cpp:2043-2079; header: model::lattice_load_bin_header, :1990-2041), called twice as the
// reference's program calls it (LOAD_state 0: the header into the run parameters, from lattice_init, :695; LOAD_state 1:
// the sample sections and the link table into the host buffers, from lattice_create_buffers, :3956), and then lets the
// reference's CPU re-measurement (SU3_CPU::SU::lattice_check_cpu, su3cpu.cpp:1616-1704; SU(2): su2cpu.cpp:1120-1210)
// measure the link table it loaded.  The reference's sources are compiled where they lie (Makefile target `ref_qcg`);
// this file fills in the public members those functions read and allocates the host buffers the reader writes to.
// No OpenCL device is touched.
//
//   qcdgpu_qcg_check file.qcg [F0MU]
//
// F0MU (0 or 1) is not in the header: the run's init file says so, for the reference as well (:2063 reads the member).
// Sizes that the reference derives from its device's workgroup size (buffer_size_align, clinterface.cpp:1580-1590;
// row size and table size, suncl.cpp:2272-2278) are derived here with 256, the value the files of this repository are
// written for.  Prints "name value" lines: every header field as loaded, the stored samples, the re-measured values.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "suncl/suncl.h"
#include "suncl/su2cpu.h"
#include "suncl/su3cpu.h"

static unsigned int align256(unsigned int n) { return (1 + (n - 1) / 256) * 256; }

int main(int argc, char** argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s file.qcg [F0MU]\n", argv[0]);
        return 2;
    }
    std::string full = argv[1];
    const size_t cut = full.find_last_of('/');
    std::string dir = cut == std::string::npos ? std::string("./") : full.substr(0, cut + 1);
    std::string name = cut == std::string::npos ? full : full.substr(cut + 1);
    model_CL::model* lat = new model_CL::model;
    char version[] = "1.5";                    // SU(N)/QCDGPU.cpp:42, copied into the model at :199-200
    lat->version = version;
    lat->path = &dir[0];
    lat->fstate = &name[0];
    lat->get_F0mu = argc > 2 && atoi(argv[2]) != 0;
    lat->LOAD_state = 0;
    lat->lattice_load_state();                 // header
    if (lat->LOAD_state != 1) {
        fprintf(stderr, "the reference's reader did not get past the header\n");
        return 1;
    }
    const int n1 = lat->lattice_full_size[0], n2 = lat->lattice_full_size[1], n3 = lat->lattice_full_size[2], n4 = lat->lattice_full_size[3];
    printf("ints %d\nrandseries %u\ngenerator %u\nnskip %u\nprng_counter %u\n", (int)lat->convert_start_to_uint(lat->ints), lat->PRNG0->PRNG_randseries,
           lat->PRNG0->convert_generator_to_uint(lat->PRNG0->PRNG_generator), (unsigned)lat->PRNG0->RL_nskip, lat->PRNG_counter);
    printf("NAV %d\nNAV_counter %d\nNITER %d\nITER %d\nITER_counter %d\nNHIT %d\nwilson_R %d\nwilson_T %d\nprecision %u\n", lat->NAV, lat->NAV_counter,
           lat->NITER, lat->ITER, lat->ITER_counter, lat->NHIT, lat->wilson_R, lat->wilson_T, lat->convert_precision_to_uint(lat->precision));
    printf("BETA %.17g\nPHI %.17g\nOMEGA %.17g\nPL_level %d\nFmunu_index1 %d\nFmunu_index2 %d\n", (double)lat->BETA, (double)lat->PHI, (double)lat->OMEGA,
           lat->PL_level, lat->Fmunu_index1, lat->Fmunu_index2);
    printf("get_Fmunu %d\nget_actions_avr %d\nget_plaquettes_avr %d\nget_wilson_loop %d\n", (int)lat->get_Fmunu, (int)lat->get_actions_avr,
           (int)lat->get_plaquettes_avr, (int)lat->get_wilson_loop);
    printf("measurement_size_F %u\nenergies_size %u\nenergies_size_F %u\npolyakov_loop_size %u\n", lat->lattice_measurement_size_F, lat->lattice_energies_size,
           lat->lattice_energies_size_F, lat->lattice_polyakov_loop_size);
    printf("group %d\nnd %d\nL %d %d %d %d\ndomain %d %d %d %d\n", lat->lattice_group, lat->lattice_nd, n1, n2, n3, n4, lat->lattice_domain_size[0],
           lat->lattice_domain_size[1], lat->lattice_domain_size[2], lat->lattice_domain_size[3]);
    if (lat->precision != model_CL::model::model_precision_double) {
        fprintf(stderr, "a single-precision table: the re-measurement below reads double4 rows\n");
        return 3;
    }

    // sizes the reference derives in model_lattice_init (suncl.cpp:2266-2278)
    const unsigned int sites = (unsigned int)(n1 * n2 * n3 * n4);
    const unsigned int rowsize = align256(sites);
    const unsigned int rows = lat->lattice_group == 3 ? 12 : 4;
    lat->lattice_table_row_size = rowsize;
    lat->lattice_table_size = align256(rowsize * rows);   // row_size * nd * elements / 4 quads
    // host buffers the second call reads into (the reference allocates them in lattice_create_buffers, :3877-3991)
    lat->plattice_measurement = (cl_double2*)calloc(lat->lattice_measurement_size_F + 1, sizeof(cl_double2));
    lat->plattice_energies = (cl_double2*)calloc(lat->lattice_energies_size + 1, sizeof(cl_double2));
    lat->plattice_energies_plq = (cl_double2*)calloc(lat->lattice_energies_size_F + 1, sizeof(cl_double2));
    lat->plattice_wilson_loop = (cl_double*)calloc(lat->lattice_energies_size + 1, sizeof(cl_double));
    lat->plattice_polyakov_loop = (cl_double2*)calloc(lat->lattice_polyakov_loop_size + 1, sizeof(cl_double2));
    lat->plattice_table_double = (cl_double4*)calloc(lat->lattice_table_size, sizeof(cl_double4));
    lat->lattice_load_state();                 // sections and link table
    if (lat->LOAD_state != 2) {
        fprintf(stderr, "the reference's reader did not read the sections\n");
        return 1;
    }
    const int ns = lat->ITER_counter < lat->ITER ? lat->ITER_counter : lat->ITER;
    const bool plq_section = lat->get_plaquettes_avr || lat->get_Fmunu || lat->get_F0mu;
    for (int i = 0; i < ns; i++) {
        printf("sample %d energies %.17g %.17g", i, lat->plattice_energies[i].s[0], lat->plattice_energies[i].s[1]);
        if (plq_section) printf(" plq %.17g %.17g", lat->plattice_energies_plq[i].s[0], lat->plattice_energies_plq[i].s[1]);
        if (lat->get_wilson_loop) printf(" wilson %.17g", lat->plattice_wilson_loop[i]);
        if (lat->PL_level > 0) printf(" polyakov %.17g %.17g", lat->plattice_polyakov_loop[i].s[0], lat->plattice_polyakov_loop[i].s[1]);
        printf("\n");
    }

    // the loaded table through the reference's CPU re-measurement (members as in ref_sun_driver.cpp)
    lat->lattice_domain_n1 = n1;
    lat->lattice_domain_n2n3 = n2 * n3;
    lat->lattice_domain_n2n3n4 = n2 * n3 * n4;
    lat->lattice_domain_n1n2n3 = n1 * n2 * n3;
    lat->lattice_full_n1n2n3 = n1 * n2 * n3;
    lat->lattice_full_site = sites;
    lat->lattice_domain_site = sites;
    lat->rowsize = rowsize;
    lat->rowsize4 = 4 * rowsize;
    const bool wilson = lat->get_wilson_loop;
    if (!wilson) lat->wilson_R = lat->wilson_T = 0;
    lat->get_plaquettes_avr = true;            // measure the plaquettes whatever the run had switched on
    lat->lattice_pointer_last = (unsigned int*)lat->plattice_table_double;
    if (lat->lattice_group == 3) {
        SU3_CPU::SU* cpu = new SU3_CPU::SU;
        cpu->GPU0 = lat->GPU0;
        cpu->lattice_check_cpu(lat);
    } else {
        SU2_CPU::SU* cpu = new SU2_CPU::SU;
        cpu->GPU0 = lat->GPU0;
        cpu->lattice_check_cpu(lat);
    }
    static const struct { const char* name; int idx; } q[] = {
        {"S_spat", DM_S_spat}, {"S_temp", DM_S_temp}, {"Plq_spat", DM_Plq_spat}, {"Plq_temp", DM_Plq_temp},
        {"Wilson_loop", DM_Wilson_loop}, {"Polyakov_loop", DM_Polyakov_loop}, {"Polyakov_loop_im", DM_Polyakov_loop_im},
        {"Polyakov_loop_P2", DM_Polyakov_loop_P2}, {"Polyakov_loop_P4", DM_Polyakov_loop_P4}};
    for (const auto& e : q) printf("check_%s %.17g\n", e.name, lat->Analysis[e.idx].CPU_last_value);
    fflush(stdout);
    _Exit(0);  // model's destructor finalises an OpenCL device: skip it
}
