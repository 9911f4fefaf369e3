// hydra_main.cpp -- `hydra_mi355x`: hydra's command line on top of the C ABI.
//
// Drop-in for `hydra --mpibayes bayesMPI --bfile X --pheno P ...`
// (src/main.cpp:17-195 -> BayesRRm::runMpiGibbs, src/BayesRRm.cpp:933): same
// flags (src/options.cpp:7-297; defaults src/options.hpp:101-127), same input
// files (.bed/.bim/.fam, .phen with NA, --groupIndexFile/--groupMixtureFile),
// same output files and byte layouts (src/BayesRRm.cpp:1071-1083,1299-1309,
// 2736-2794): <dir>/<name>.csv .bet .cpn .acu .mus.<rank>, and on --save
// .eps.<rank> .mrk.<rank> .rng.<rank> .xbet .xcpn (:2802-2838); --restart
// [--ignore-xfiles] resumes from those dumps into <name>_rs.* (:842-928,
// :1197-1220).
//
// `--mpibayes bayesWMPI --failure F --quad_points Q` runs BayesW (src/BayesW.cpp:905), sharded the same way.
//
// Not reproduced (out of scope for the hot path): sparse
// file formats, bayesFH, marker-sharded MPI, the .lst/tarball.
// Multi-GPU: one process per GPU (RANK/WORLD_SIZE/LOCAL_RANK in the
// environment, as torchrun/mpirun export them); individuals are sharded and the
// ncclUniqueId travels through a file in --mcmc-out-dir.
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hgibbs.h"

namespace {

struct Options { // src/options.hpp:20-138 (subset that reaches bayesMPI)
    std::string bayesType, analysisType = "Bayes";
    std::string bedFile, phenotypeFile, mcmcOutDir, mcmcOutNam, groupIndexFile, groupMixtureFile, covariatesFile;
    bool covariates = false;
    unsigned chainLength = 10000, burnin = 5000, thin = 5, save = 10;
    unsigned seed = 0;
    bool seedGiven = false;
    unsigned numberMarkers = 0, numberIndividuals = 0;
    int shuffleMarkers = 1, syncRate = 1;
    std::vector<double> S{0.01, 0.001, 0.0001};
    bool readFromBedFile = false;
    bool restart = false, useXfilesInRestart = true; // options.hpp:33-34
    std::string failureFile, quad_points;            // options.hpp:56-58 (bayesWMPI)
    int batch = 0, cpg = 0; // tuning knobs of this build (not hydra's)
};

[[noreturn]] void fatal(const std::string& m)
{
    std::fprintf(stderr, "\n%s\n", m.c_str());
    std::exit(1);
}

// Gadget::Tokenizer::getTokens, src/gadgets.cpp:12-22
std::vector<std::string> tokens(const std::string& str, const std::string& sep)
{
    std::vector<std::string> out;
    std::string::size_type b = str.find_first_not_of(sep);
    while (b != std::string::npos) {
        std::string::size_type e = str.find_first_of(sep, b);
        if (e == std::string::npos) e = str.length();
        out.push_back(str.substr(b, e - b));
        b = str.find_first_not_of(sep, e);
    }
    return out;
}

// Options::readFile, src/options.cpp:335-397: "key value" pairs; a key that starts with // or # skips its value token; an
// unknown key is an error.  mcmcOut is the output prefix (directory/name); the bed file named here is read as --bfile would.
void read_option_file(const std::string& file, Options& o)
{
    std::ifstream in(file.c_str());
    if (!in) fatal("Error: can not open the file [" + file + "] to read.");
    std::string key, value;
    while (in >> key >> value) {
        if (key == "bedFile") {
            o.bedFile = value;
            o.readFromBedFile = true;
        } else if (key == "phenotypeFile") o.phenotypeFile = value;
        else if (key == "analysisType") o.analysisType = value;
        else if (key == "bayesType") o.bayesType = value;
        else if (key == "mcmcOut") {
            const std::string::size_type cut = value.find_last_of('/');
            o.mcmcOutDir = cut == std::string::npos ? std::string(".") : (cut == 0 ? std::string("/") : value.substr(0, cut));
            o.mcmcOutNam = cut == std::string::npos ? value : value.substr(cut + 1);
        } else if (key == "shuffleMarkers") o.shuffleMarkers = std::stoi(value);
        else if (key == "syncRate") o.syncRate = std::stoi(value);
        else if (key == "blocksPerRank") (void)std::stoi(value); // hydra's marker-sharded MPI layout: no meaning here (individuals shard)
        else if (key == "numberMarkers") o.numberMarkers = (unsigned)std::stoi(value);
        else if (key == "numberIndividuals") o.numberIndividuals = (unsigned)std::stoi(value);
        else if (key == "chainLength") o.chainLength = (unsigned)std::stoi(value);
        else if (key == "burnin") o.burnin = (unsigned)std::stoi(value);
        else if (key == "seed") {
            o.seed = (unsigned)std::stoi(value);
            o.seedGiven = true;
        } else if (key == "thin") o.thin = (unsigned)std::stoi(value);
        else if (key == "save") o.save = (unsigned)std::stoi(value);
        else if (key == "S") {
            o.S.clear();
            // (the reference reads these through stof, src/options.cpp:384: the option file's 0.0001 is 9.99999974737875e-05 in the chain,
            // unlike --S on the command line, which goes through stod)
            for (const std::string& t : tokens(value, " ,")) o.S.push_back((double)std::stof(t));
        } else if (key.substr(0, 2) == "//" || key.substr(0, 1) == "#") {
            continue;
        } else {
            fatal("\nError: invalid option " + key + " " + value + "\n");
        }
    }
}

Options parse(int argc, const char* argv[])
{
    Options o;
    auto need = [&](int& i) -> const char* {
        if (i + 1 >= argc) fatal(std::string("missing value after ") + argv[i]);
        return argv[++i];
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--inp-file") { // options.cpp:8-11: the file replaces the rest of the command line
            read_option_file(need(i), o);
            if (!o.seedGiven) o.seed = (unsigned)std::time(nullptr);
            return o;
        }
        if (a == "--mpibayes" || a == "--bayesType") { // --bayesType: alias (it is the option-file key, options.cpp:353)
            o.analysisType = "RAM";
            o.bayesType = need(i);
        } else if (a == "--bfile") {
            o.readFromBedFile = true;
            o.bedFile = need(i);
        } else if (a == "--pheno") {
            o.phenotypeFile = tokens(need(i), ",").at(0);
        } else if (a == "--mcmc-out-dir") o.mcmcOutDir = need(i);
        else if (a == "--mcmc-out-name") o.mcmcOutNam = need(i);
        else if (a == "--shuf-mark") o.shuffleMarkers = std::atoi(need(i));
        else if (a == "--sync-rate") o.syncRate = std::atoi(need(i));
        else if (a == "--number-markers") o.numberMarkers = (unsigned)std::atoi(need(i));
        else if (a == "--number-individuals") o.numberIndividuals = (unsigned)std::atoi(need(i));
        else if (a == "--chain-length") o.chainLength = (unsigned)std::atoi(need(i));
        else if (a == "--burn-in") o.burnin = (unsigned)std::atoi(need(i));
        else if (a == "--seed") {
            o.seed = (unsigned)std::atoi(need(i));
            o.seedGiven = true;
        } else if (a == "--thin") o.thin = (unsigned)std::atoi(need(i));
        else if (a == "--save") o.save = (unsigned)std::atoi(need(i));
        else if (a == "--S") {
            o.S.clear();
            for (const std::string& t : tokens(need(i), " ,")) o.S.push_back(std::stod(t));
        } else if (a == "--groupIndexFile") o.groupIndexFile = need(i);
        else if (a == "--groupMixtureFile") o.groupMixtureFile = need(i);
        else if (a == "--batch") o.batch = std::atoi(need(i));
        else if (a == "--cols-per-group") o.cpg = std::atoi(need(i));
        else if (a == "--covariates") {
            o.covariates = true;
            o.covariatesFile = need(i);
        } else if (a == "--failure") o.failureFile = need(i);    // options.cpp:183-186
        else if (a == "--quad_points") o.quad_points = need(i); // options.cpp:188-191
        else if (a == "--restart") o.restart = true;          // options.cpp:63-65
        else if (a == "--ignore-xfiles") o.useXfilesInRestart = false; // options.cpp:67-69
        else if (a == "--sparse-dir" || a == "--sparse-basename" ||
                 a == "--bed-to-sparse" || a == "--sparse-sync" || a == "--bed-sync")
            fatal("FATAL  : option " + a + " belongs to a part of hydra this build does not reproduce (SURVEY.md section 2)");
        else
            fatal("\nError: invalid option \"" + a + "\".\n"); // options.cpp:292-295
    }
    if (!o.seedGiven) o.seed = (unsigned)std::time(nullptr); // options.hpp:105
    if (o.analysisType == "RAM") {                           // options.cpp:303-326
        if (o.mcmcOutDir.empty()) fatal("FATAL  : --mcmc-out-dir is mandatory with --mpibayes");
        if (o.mcmcOutNam.empty()) fatal("FATAL  : --mcmc-out-name is mandatory with --mpibayes");
    }
    return o;
}

size_t count_fam(const std::string& path, std::vector<std::string>* ids)
{
    std::ifstream in(path);
    if (!in) fatal("Error: can not open the file [" + path + "] to read.");
    std::string fid, pid, dad, mom, sex, phen;
    size_t n = 0;
    std::map<std::string, int> seen;
    while (in >> fid >> pid >> dad >> mom >> sex >> phen) { // data.cpp:1454
        const std::string id = fid + ":" + pid;
        if (!seen.emplace(id, 1).second) fatal("Error: Duplicate individual ID found: \"" + fid + "\t" + pid + "\".");
        if (ids) ids->push_back(id);
        ++n;
    }
    return n;
}

size_t count_bim(const std::string& path)
{
    std::ifstream in(path);
    if (!in) fatal("Error: can not open the file [" + path + "] to read.");
    std::string id, a1, a2;
    unsigned chr, pos;
    float gpos;
    size_t n = 0;
    while (in >> chr >> id >> gpos >> pos >> a1 >> a2) ++n; // data.cpp:1484
    return n;
}

// Data::readPhenotypeFile(phenFile), src/data.cpp:1840-1882: lines whose FID:IID
// is in the .fam are taken in FILE order; "NA" marks a dropped individual.
void read_phen(const std::string& path, const std::vector<std::string>& fam_ids, std::vector<double>& y,
               std::vector<uint8_t>& keep)
{
    std::ifstream in(path);
    if (!in) fatal("Error: can not open the phenotype file [" + path + "] to read.");
    std::map<std::string, int> idx;
    for (size_t i = 0; i < fam_ids.size(); ++i) idx[fam_ids[i]] = (int)i;
    keep.assign(fam_ids.size(), 1);
    y.clear();
    std::string line;
    size_t lineno = 0;
    while (std::getline(in, line)) {
        std::vector<std::string> col = tokens(line, " \t");
        if (col.size() < 3) continue;
        if (!idx.count(col[0] + ":" + col[1])) continue;
        if (lineno >= keep.size()) break;
        if (col[2] != "NA") y.push_back(std::atof(col[2].c_str()));
        else keep[lineno] = 0; // NAsInds.push_back(line)
        ++lineno;
    }
    if (lineno != fam_ids.size()) fatal("FATAL  : phenotype file covers " + std::to_string(lineno) + " of " + std::to_string(fam_ids.size()) + " individuals");
}

// Data::readPhenCovFiles, src/data.cpp:1615-1673: .phen and .cov are read line by line in
// lockstep (no .fam lookup); an individual is dropped if its phenotype or any covariate is NA.
void read_phen_cov(const std::string& phen, const std::string& cov, size_t numInds, std::vector<double>& y,
                   std::vector<uint8_t>& keep, std::vector<double>& X, int& C)
{
    std::ifstream inp(phen), inc(cov);
    if (!inp) fatal("Error: can not open the phenotype file [" + phen + "] to read.");
    if (!inc) fatal("Error: can not open the covariates file [" + cov + "] to read.");
    keep.assign(numInds, 1);
    y.clear();
    X.clear();
    std::string lp, lc;
    size_t line = 0;
    while (std::getline(inp, lp)) {
        if (!std::getline(inc, lc)) fatal("FATAL  : covariates file is shorter than the phenotype file");
        if (line >= numInds) break;
        std::vector<std::string> cp = tokens(lp, " \t"), cc = tokens(lc, " \t");
        if (cp.size() < 3) continue;
        bool naC = false;
        for (size_t i = 2; i < cc.size(); ++i)
            if (cc[i] == "NA") naC = true;
        if (cp[2] != "NA" && !naC) {
            y.push_back(std::atof(cp[2].c_str()));
            for (size_t i = 2; i < cc.size(); ++i) X.push_back(std::stod(cc[i]));
        } else {
            keep[line] = 0;
        }
        ++line;
    }
    if (line != numInds) fatal("FATAL  : phenotype/covariates files cover " + std::to_string(line) + " of " + std::to_string(numInds) + " individuals");
    C = y.empty() ? 0 : (int)(X.size() / y.size());
    std::printf("numFixedEffect = %d\n", C);
}

std::vector<int32_t> read_groups(const std::string& path) // data.cpp:1940-1958
{
    std::ifstream in(path);
    if (!in) fatal("Error: can not open the group file [" + path + "] to read. Use the --groupIndexFile option!");
    std::vector<int32_t> g;
    int v;
    while (in >> v) g.push_back(v);
    return g;
}

std::vector<std::vector<double>> read_mS(const std::string& path) // data.cpp:1963-2004
{
    std::ifstream in(path);
    if (!in) fatal("Error: can not open the mixture file [" + path + "] to read. Use the --groupMixtureFile option!");
    std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<std::string> rows = tokens(text, ";");
    std::vector<std::vector<double>> mS;
    size_t ncomp = 0;
    for (const std::string& r : rows) {
        std::vector<std::string> t = tokens(r, ",");
        // a trailing newline after the last ';' group is not a group
        bool blank = true;
        for (char ch : r)
            if (!std::isspace((unsigned char)ch)) blank = false;
        if (blank) continue;
        if (ncomp == 0) ncomp = t.size();
        if (t.size() != ncomp) fatal("FATAL  : all group mixture should have the same number of components");
        std::vector<double> row{0.0};
        for (const std::string& x : t) {
            const double mix = std::stod(x);
            if (mix <= 0.0) fatal("FATAL  : mixture value can only be strictly positive");
            row.push_back(mix);
        }
        mS.push_back(row);
    }
    return mS;
}

void hg_check(int rc, const char* what)
{
    if (rc) fatal(std::string("FATAL  : ") + what + ": " + hgibbs_last_error());
}

void pwrite_at(FILE* f, long off, const void* p, size_t n)
{
    if (std::fseek(f, off, SEEK_SET) != 0 || std::fwrite(p, 1, n, f) != n) fatal("FATAL  : short write on an output file");
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---- restart readers: Data::read_mcmc_output_*_file, src/data.cpp:33-519 ----
struct CsvRestart {
    unsigned iteration_to_restart_from = 0, first_thinned_iteration = 0, first_saved_iteration = 0;
    std::vector<double> sigmaG, pi;
    double sigmaE = 0.0;
};

// data.cpp:406-514: the last line whose iteration is a multiple of --save wins
CsvRestart read_csv_for_restart(const std::string& csv, unsigned thin, unsigned save, int G, int K)
{
    std::ifstream file(csv);
    if (!file) fatal("*FATAL*: failed to open csv file " + csv + "!");
    CsvRestart r;
    r.sigmaG.assign(G, 0.0);
    r.pi.assign((size_t)G * K, 0.0);
    int nSaved = 0, nThinned = 0, last_it = -1;
    std::string str;
    while (std::getline(file, str)) {
        if (str.empty()) continue;
        for (char& ch : str)
            if (ch == ',') ch = ' ';
        char* p = &str[0];
        const long it = std::strtol(p, &p, 10);
        const long ngrp = std::strtol(p, &p, 10);
        last_it = (int)it;
        if (it % thin != 0) fatal("FATAL  : " + csv + ": iteration " + std::to_string(it) + " is not a multiple of --thin");
        if (++nThinned == 1) r.first_thinned_iteration = (unsigned)it;
        if (it % save != 0) continue;
        if (++nSaved == 1) r.first_saved_iteration = (unsigned)it;
        r.iteration_to_restart_from = (unsigned)it;
        if (ngrp != G) fatal("FATAL  : " + csv + ": number of groups differs from this run's");
        for (int g = 0; g < G; ++g) r.sigmaG[g] = std::strtod(p, &p);
        r.sigmaE = std::strtod(p, &p);
        (void)std::strtod(p, &p);      // sigmaG.sum()/(sigmaE+sigmaG.sum())
        (void)std::strtol(p, &p, 10);  // m0
        const long rows = std::strtol(p, &p, 10), cols = std::strtol(p, &p, 10);
        if (rows != G || cols != K) fatal("FATAL  : " + csv + ": pi is " + std::to_string(rows) + "x" + std::to_string(cols) + ", expected " + std::to_string(G) + "x" + std::to_string(K));
        for (size_t i = 0; i < (size_t)G * K; ++i) r.pi[i] = std::strtod(p, &p);
    }
    if (nSaved == 0)
        fatal("FATAL  : No saved iteration could be found when reading " + csv + "!\n       : with last read iteration " + std::to_string(last_it) +
              ", and --thin = " + std::to_string(thin) + " and --save = " + std::to_string(save));
    return r;
}

void pread_at(FILE* f, long off, void* dst, size_t n, const std::string& what)
{
    if (std::fseek(f, off, SEEK_SET) != 0 || std::fread(dst, 1, n, f) != n) fatal("FATAL  : short read from " + what);
}

FILE* open_ro(const std::string& p)
{
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) fatal("FATAL  : can not open " + p + " to restart from");
    return f;
}

void expect_u(unsigned got, unsigned want, const std::string& what)
{
    if (got != want) fatal("Mismatch between expected and read " + what + ": " + std::to_string(want) + " vs " + std::to_string(got));
}

// .bet/.cpn history (elem = 8/4 bytes) or the single-line .xbet/.xcpn, data.cpp:256-402
void read_marker_history(const std::string& path, bool xfile, unsigned Mtot, unsigned it_from, unsigned first_thinned, unsigned thin, size_t elem,
                         void* dst)
{
    FILE* f = open_ro(path);
    unsigned Mtot_ = 0, it_ = ~0u;
    pread_at(f, 0, &Mtot_, sizeof(unsigned), path);
    expect_u(Mtot_, Mtot, path + " Mtot");
    if ((it_from - first_thinned) % thin != 0) fatal("FATAL  : " + path + ": restart iteration is not on the --thin grid");
    const long n_skip = (long)((it_from - first_thinned) / thin);
    const long off = xfile ? (long)sizeof(unsigned) : (long)sizeof(unsigned) + n_skip * (long)(sizeof(unsigned) + (size_t)Mtot * elem);
    pread_at(f, off, &it_, sizeof(unsigned), path);
    expect_u(it_, it_from, path + " iteration");
    pread_at(f, off + (long)sizeof(unsigned), dst, (size_t)Mtot * elem, path);
    std::fclose(f);
}

// .eps/.mrk/.gam/.xiv dumps: (iteration, length, payload), data.cpp:33-204
std::vector<uint8_t> read_dump(const std::string& path, unsigned it_from, size_t elem, unsigned* length)
{
    FILE* f = open_ro(path);
    unsigned it_ = ~0u, len = 0;
    pread_at(f, 0, &it_, sizeof(unsigned), path);
    expect_u(it_, it_from, path + " iteration");
    pread_at(f, sizeof(unsigned), &len, sizeof(unsigned), path);
    std::vector<uint8_t> out((size_t)len * elem);
    pread_at(f, 2 * sizeof(unsigned), out.data(), out.size(), path);
    std::fclose(f);
    *length = len;
    return out;
}

// `file >> rng` for boost::mt19937, src/distributions_boost.cpp:46-55: 624 decimal words
void read_rng_file(const std::string& path, hgibbs_rng_state* st)
{
    std::ifstream in(path);
    if (!in) fatal("*FATAL*: Unable to read from file " + path);
    std::vector<uint32_t> w(624);
    for (int i = 0; i < 624; ++i) {
        unsigned long long v = 0;
        if (!(in >> v) || v > 0xffffffffull) fatal("*FATAL*: " + path + " does not hold a boost::mt19937 state");
        w[i] = (uint32_t)v;
    }
    hydra_rng_from_boost_words(w.data(), st);
}

// `file << rng`, src/distributions_boost.cpp:38-44: words separated by single spaces
void write_rng_file(const std::string& path, const hgibbs_rng_state& st)
{
    std::vector<uint32_t> w(624);
    hydra_rng_to_boost_words(&st, w.data());
    std::ofstream out(path, std::ios::out | std::ios::trunc | std::ios::binary);
    if (!out) fatal("FATAL  : can not create " + path);
    for (int i = 0; i < 624; ++i) out << w[i] << (i + 1 < 624 ? " " : "");
}

// Several ranks (one process per GPU, RANK / WORLD_SIZE / LOCAL_RANK in the environment): the RCCL id and the
// IPC handles of the in-launch peer mailboxes travel through files next to the outputs.
void setup_ranks(hgibbs_t dev, const std::string& base, int rank, int nranks)
{
        uint8_t id[128];
        const std::string idf = base + ".ncclid";
        if (rank == 0) {
            hg_check(hgibbs_comm_unique_id(id), "hgibbs_comm_unique_id");
            FILE* f = std::fopen((idf + ".tmp").c_str(), "wb");
            if (!f || std::fwrite(id, 1, 128, f) != 128) fatal("FATAL  : can not write " + idf);
            std::fclose(f);
            std::rename((idf + ".tmp").c_str(), idf.c_str());
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = std::fopen(idf.c_str(), "rb")); ++tries)
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            if (!f || std::fread(id, 1, 128, f) != 128) fatal("FATAL  : can not read " + idf);
            std::fclose(f);
        }
        hg_check(hgibbs_comm_init(dev, nranks, rank, id), "hgibbs_comm_init");
        // in-launch peer-mailbox exchange: every rank publishes its IPC handle next to the id file
        uint8_t mine[64];
        std::vector<uint8_t> all((size_t)nranks * 64);
        bool p2p_ok = hgibbs_p2p_export(dev, mine) == 0;
        {
            const std::string hf = base + ".p2p." + std::to_string(rank);
            FILE* f = std::fopen((hf + ".tmp").c_str(), "wb");
            if (f) {
                std::fwrite(p2p_ok ? "Y" : "N", 1, 1, f);
                std::fwrite(mine, 1, 64, f);
                std::fclose(f);
                std::rename((hf + ".tmp").c_str(), hf.c_str());
            }
        }
        for (int r = 0; r < nranks; ++r) {
            const std::string hf = base + ".p2p." + std::to_string(r);
            FILE* f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = std::fopen(hf.c_str(), "rb")); ++tries)
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            char flag = 'N';
            if (!f || std::fread(&flag, 1, 1, f) != 1 || std::fread(all.data() + (size_t)r * 64, 1, 64, f) != 64 || flag != 'Y') p2p_ok = false;
            if (f) std::fclose(f);
        }
        if (p2p_ok && hgibbs_p2p_import(dev, all.data()) != 0) p2p_ok = false;
        // a rank that could not import falls back to RCCL; all ranks must agree, so publish the verdict too
        {
            const std::string vf = base + ".p2pok." + std::to_string(rank);
            FILE* f = std::fopen((vf + ".tmp").c_str(), "wb");
            if (f) {
                std::fwrite(p2p_ok ? "Y" : "N", 1, 1, f);
                std::fclose(f);
                std::rename((vf + ".tmp").c_str(), vf.c_str());
            }
            for (int r = 0; r < nranks; ++r) {
                const std::string of = base + ".p2pok." + std::to_string(r);
                FILE* g = nullptr;
                for (int tries = 0; tries < 6000 && !(g = std::fopen(of.c_str(), "rb")); ++tries)
                    std::this_thread::sleep_for(std::chrono::milliseconds(10));
                char flag = 'N';
                if (!g || std::fread(&flag, 1, 1, g) != 1 || flag != 'Y') p2p_ok = false;
                if (g) std::fclose(g);
            }
        }
        hg_check(hgibbs_set_option(dev, "p2p", p2p_ok ? 1 : 0), "p2p");
        if (rank == 0) std::printf("INFO   : per-batch exchange over %s\n", p2p_ok ? "xGMI peer mailboxes (in-launch)" : "RCCL all-reduce");
        if (rank == 0) {
            std::this_thread::sleep_for(std::chrono::milliseconds(500));
            std::remove(idf.c_str());
            for (int r = 0; r < nranks; ++r) {
                std::remove((base + ".p2p." + std::to_string(r)).c_str());
                std::remove((base + ".p2pok." + std::to_string(r)).c_str());
            }
        }
}

// Data::readPhenFailFiles / readPhenFailCovFiles, src/data.cpp:1681-1802: .phen, .fail (and .cov)
// are read line by line in lockstep; an individual is dropped if its phenotype is NA, its failure
// indicator is -9, or any covariate is NA.
void read_phen_fail(const std::string& phen, const std::string& failf, const std::string& cov, size_t numInds, std::vector<double>& y,
                    std::vector<int32_t>& fail, std::vector<uint8_t>& keep, std::vector<double>& X, int& C)
{
    std::ifstream inp(phen), inf(failf), inc;
    if (!inp) fatal("Error: can not open the phenotype file [" + phen + "] to read.");
    if (!inf) fatal("Error: can not open the failure file [" + failf + "] to read.");
    if (!cov.empty()) {
        inc.open(cov);
        if (!inc) fatal("Error: can not open the covariates file [" + cov + "] to read.");
    }
    keep.assign(numInds, 1);
    y.clear();
    fail.clear();
    X.clear();
    std::string lp, lf, lc;
    size_t line = 0;
    while (std::getline(inp, lp)) {
        if (!std::getline(inf, lf)) fatal("FATAL  : failure file is shorter than the phenotype file");
        if (!cov.empty() && !std::getline(inc, lc)) fatal("FATAL  : covariates file is shorter than the phenotype file");
        if (line >= numInds) break;
        std::vector<std::string> cp = tokens(lp, " \t"), cf = tokens(lf, " \t"), cc = tokens(lc, " \t");
        if (cp.size() < 3 || cf.empty()) continue;
        bool naC = false;
        for (size_t i = 2; i < cc.size(); ++i)
            if (cc[i] == "NA") naC = true;
        if (cp[2] != "NA" && !naC && cf[0] != "-9") {
            y.push_back(std::atof(cp[2].c_str()));
            const double d = std::atof(cf[0].c_str());
            if (d != 0.0 && d != 1.0) fatal("FATAL  : failure indicator on line " + std::to_string(line) + " is neither 0, 1 nor -9");
            fail.push_back((int32_t)d);
            for (size_t i = 2; i < cc.size(); ++i) X.push_back(std::stod(cc[i]));
        } else {
            std::cout << "NA(s) detected on line " << line << ", naP? " << cp[2] << ", naF? " << cf[0] << std::endl;
            keep[line] = 0;
        }
        ++line;
    }
    if (line != numInds) fatal("FATAL  : phenotype/failure files cover " + std::to_string(line) + " of " + std::to_string(numInds) + " individuals");
    C = (y.empty() || cov.empty()) ? 0 : (int)(X.size() / y.size());
}

// --mpibayes bayesWMPI: BayesW::runMpiGibbs_bW, src/BayesW.cpp:905-2176
int run_bayesw(const Options& opt_in, int rank, int nranks, int local_rank)
{
    Options opt = opt_in;
    if (opt.failureFile.empty()) fatal("FATAL  : --failure is mandatory with --mpibayes bayesWMPI");
    if (opt.quad_points.empty()) fatal("Possible number of quad_points = 3,5,7,9,11,13,15,17,25"); // src/BayesW.cpp:706-708
    const int quad = std::atoi(opt.quad_points.c_str());
    std::vector<std::string> fam_ids;
    const size_t numInds = count_fam(opt.bedFile + ".fam", &fam_ids);
    const size_t numSnps = count_bim(opt.bedFile + ".bim");
    std::vector<double> y, covX;
    std::vector<int32_t> fail;
    std::vector<uint8_t> keep;
    int C = 0;
    read_phen_fail(opt.phenotypeFile, opt.failureFile, opt.covariates ? opt.covariatesFile : std::string(), numInds, y, fail, keep, covX, C);
    const unsigned numNAs = (unsigned)(numInds - y.size());
    if (opt.numberIndividuals == 0) fatal("FATAL  : opt.numberIndividuals is zero! Set it via --number-individuals in call.");
    if (opt.numberMarkers == 0) fatal("FATAL  : opt.numberMarkers is zero! Set it via --number-markers in call.");
    if (opt.numberIndividuals != numInds) fatal("FATAL  : --number-individuals does not match the .fam file");
    unsigned Mtot = opt.numberMarkers;
    if (Mtot > numSnps) fatal("FATAL  : --number-markers exceeds the .bim file");
    const unsigned Ntot = (unsigned)numInds - numNAs;
    std::printf("INFO   : Full dataset includes Mtot=%d markers and Ntot=%d individuals.\n", Mtot, (int)numInds);

    std::vector<int32_t> groups;
    std::vector<std::vector<double>> mS;
    if (!opt.groupIndexFile.empty()) { // src/BayesW.cpp:773-776
        groups = read_groups(opt.groupIndexFile);
        mS = read_mS(opt.groupMixtureFile);
        if (groups.size() < Mtot) fatal("FATAL  : group file covers fewer markers than --number-markers");
        groups.resize(Mtot);
    } else {
        std::vector<double> row{0.0};
        for (double v : opt.S) row.push_back(v);
        mS.push_back(row);
    }
    const int G = (int)mS.size(), K = (int)mS[0].size();
    std::printf("numGroups = %d, data.groups.size() = %lu, Mtot = %d\n", G, (unsigned long)Mtot, Mtot); // :778
    std::vector<double> mS_flat;
    for (auto& r : mS) mS_flat.insert(mS_flat.end(), r.begin(), r.end());
    if (opt.save < opt.thin) { // :981-989
        opt.save = opt.thin;
        std::printf("WARNING: opt.save was lower that opt.thin ; opt.save reset to opt.thin (%d)\n", opt.thin);
    }
    if (opt.save % opt.thin != 0) {
        std::printf("WARNING: opt.save (= %d) was not a multiple of opt.thin (= %d)\n", opt.save, opt.thin);
        opt.save = (opt.save / opt.thin) * opt.thin;
        std::printf("         opt.save reset to %d, the closest multiple of opt.thin (%d)\n", opt.save, opt.thin);
    }
    struct stat sb;
    if (stat(opt.mcmcOutDir.c_str(), &sb) != 0)
        if (std::system(("mkdir -p " + opt.mcmcOutDir).c_str()) != 0) fatal("FATAL  : can not create --mcmc-out-dir");
    const std::string base_in = opt.mcmcOutDir + "/" + opt.mcmcOutNam; // a restart reads <name>.*, writes <name>_rs.* (:994-1008)
    const std::string base = opt.restart ? base_in + "_rs" : base_in;

    hgibbs_t dev = nullptr;
    hg_check(hgibbs_create(local_rank, &dev), "hgibbs_create");
    if (nranks > 1) setup_ranks(dev, base, rank, nranks);
    if (opt.batch) hg_check(hgibbs_set_option(dev, "batch", opt.batch), "batch");
    const double tl0 = now_s();
    const size_t snpLenByt = (numInds + 3) / 4;
    // individuals sharded in multiples of 4 of the KEPT rows
    const unsigned per = ((Ntot + nranks - 1) / nranks + 3) / 4 * 4;
    const unsigned lo = std::min(Ntot, rank * per), hi = std::min(Ntot, (rank + 1) * per);
    {
        std::vector<uint8_t> bed;
        std::ifstream in(opt.bedFile + ".bed", std::ios::binary);
        if (!in) fatal("Error: can not open the file [" + opt.bedFile + ".bed] to read.");
        unsigned char magic[3];
        in.read((char*)magic, 3);
        if (!in || magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01) fatal("FATAL  : " + opt.bedFile + ".bed is not a SNP-major PLINK bed");
        bed.resize((size_t)Mtot * snpLenByt);
        in.read((char*)bed.data(), (std::streamsize)bed.size());
        if ((size_t)in.gcount() != bed.size()) fatal("FATAL  : " + opt.bedFile + ".bed is shorter than M x ceil(N/4)");
        hg_check(hgibbs_load_bed(dev, bed.data(), snpLenByt, (uint32_t)numInds, Mtot, numNAs ? keep.data() : nullptr, lo, hi, Ntot), "hgibbs_load_bed");
        std::printf("INFO   : rank %3d took %.3f seconds to load  %lu bytes  =>  BW = %7.3f GB/s\n", rank, now_s() - tl0, (unsigned long)bed.size(),
                    (double)bed.size() * 1e-9 / (now_s() - tl0));
    }
    if (numNAs) std::printf("INFO   : Ntot adjusted by -%d to account for NAs in phenotype file. Now Ntot=%d\n", numNAs, Ntot);

    hydraw_model_desc md{};
    md.seed = opt.seed;
    md.shuffle = opt.shuffleMarkers;
    md.G = G;
    md.K = K;
    md.groups = groups.empty() ? nullptr : groups.data();
    md.mS = mS_flat.data();
    md.quad_points = quad;
    hydraw_chain_t chain = nullptr;
    hg_check(hydraw_chain_create(dev, &md, y.data(), fail.data(), &chain), "hydraw_chain_create");
    if (opt.covariates) hg_check(hydraw_chain_set_covariates(chain, covX.data(), C), "hydraw_chain_set_covariates");

    // ---- --restart: BayesW::init_from_restart, src/BayesW.cpp:869-903 (readers src/data.cpp:523-663) ----
    unsigned iteration_start = 0;
    if (opt.restart) {
        std::printf("RESTART: from files: %s.* files\n", base_in.c_str());
        std::ifstream file(base_in + ".csv");
        if (!file) fatal("*FATAL*: failed to open csv file " + base_in + ".csv!");
        unsigned it_from = 0, first_thinned = 0;
        int nSaved = 0, nThinned = 0;
        double r_mu = 0.0, r_alpha = 0.0;
        std::vector<double> r_sigmaG(G), r_pi((size_t)G * K);
        std::string str;
        while (std::getline(file, str)) {
            if (str.empty()) continue;
            for (char& ch : str)
                if (ch == ',') ch = ' ';
            char* q = &str[0];
            const long it = std::strtol(q, &q, 10);
            if (it % opt.thin != 0) fatal("FATAL  : " + base_in + ".csv: iteration " + std::to_string(it) + " is not a multiple of --thin");
            if (++nThinned == 1) first_thinned = (unsigned)it;
            if (it % opt.save != 0) continue;
            ++nSaved;
            it_from = (unsigned)it;
            r_mu = std::strtod(q, &q);
            (void)std::strtod(q, &q); // sigmaG.sum()
            r_alpha = std::strtod(q, &q);
            (void)std::strtod(q, &q); // h2
            (void)std::strtol(q, &q, 10); // m0
            const long rows = std::strtol(q, &q, 10), cols = std::strtol(q, &q, 10);
            if (rows != G || cols != K) fatal("FATAL  : " + base_in + ".csv: pi is " + std::to_string(rows) + "x" + std::to_string(cols) + ", expected " + std::to_string(G) + "x" + std::to_string(K));
            for (int g = 0; g < G; ++g) r_sigmaG[g] = std::strtod(q, &q);
            for (size_t i = 0; i < (size_t)G * K; ++i) r_pi[i] = std::strtod(q, &q);
        }
        if (nSaved == 0) fatal("FATAL  : No saved iteration could be found when reading " + base_in + ".csv!");
        if (it_from == 0) fatal("FATAL  : There is no point in restarting a chain from iteration 0 (not saved anyway)\n         => restart your analysis from scratch");
        const bool xf = opt.useXfilesInRestart;
        std::vector<double> r_beta(Mtot);
        std::vector<int32_t> r_comp(Mtot);
        read_marker_history(base_in + (xf ? ".xbet" : ".bet"), xf, Mtot, it_from, first_thinned, opt.thin, sizeof(double), r_beta.data());
        read_marker_history(base_in + (xf ? ".xcpn" : ".cpn"), xf, Mtot, it_from, first_thinned, opt.thin, sizeof(int32_t), r_comp.data());
        unsigned len = 0;
        std::vector<uint8_t> eb = read_dump(base_in + ".eps." + std::to_string(rank), it_from, sizeof(double), &len);
        const double* r_eps = (const double*)eb.data();
        if (len == Ntot && len != hi - lo) r_eps += lo; // a full-length dump is sliced
        else expect_u(len, hi - lo, ".eps Ntot");
        std::vector<uint8_t> mb = read_dump(base_in + ".mrk." + std::to_string(rank), it_from, sizeof(int32_t), &len);
        expect_u(len, Mtot, ".mrk M");
        std::vector<double> r_gamma(C);
        std::vector<uint8_t> xb;
        if (opt.covariates) { // text .gam: the last line whose iteration is a multiple of --save (src/data.cpp:624-663)
            std::ifstream gf(base_in + ".gam");
            if (!gf) fatal("*FATAL*: failed to open csv file " + base_in + ".gam!");
            long g_it = -1;
            while (std::getline(gf, str)) {
                if (str.empty()) continue;
                for (char& ch : str)
                    if (ch == ',') ch = ' ';
                char* q = &str[0];
                const long it = std::strtol(q, &q, 10);
                if (it % opt.save != 0) continue;
                g_it = it;
                for (int i = 0; i < C; ++i) r_gamma[i] = std::strtod(q, &q);
            }
            expect_u((unsigned)g_it, it_from, ".gam iteration");
            xb = read_dump(base_in + ".xiv", it_from, sizeof(int32_t), &len);
            expect_u(len, (unsigned)C, ".xiv length");
        }
        hydraw_restart_state rs{};
        rs.iteration = it_from;
        rs.mu = r_mu;
        rs.alpha = r_alpha;
        rs.sigmaG = r_sigmaG.data();
        rs.pi = r_pi.data();
        rs.beta = r_beta.data();
        rs.components = r_comp.data();
        rs.eps = r_eps;
        rs.order = (const int32_t*)mb.data();
        rs.gamma = opt.covariates ? r_gamma.data() : nullptr;
        rs.xI = opt.covariates ? (const int32_t*)xb.data() : nullptr;
        rs.ars_seed = opt.seed + it_from; // srand(opt.seed + iteration_to_restart_from), :877
        read_rng_file(base_in + ".rng." + std::to_string(rank), &rs.rng);
        hg_check(hydraw_chain_restore(chain, &rs), "hydraw_chain_restore");
        iteration_start = it_from + 1;
        std::printf("INFO   : %s\nINFO   : RESTART DETECTED\nINFO   : restarting from: %s.* files\n", std::string(100, '*').c_str(), base_in.c_str());
        std::printf("INFO   : last saved iteration:        %d\nINFO   : will restart from iteration: %d\nINFO   : %s\n", it_from, iteration_start,
                    std::string(100, '*').c_str());
    }

    auto open_trunc = [&](const std::string& p) {
        FILE* f = std::fopen(p.c_str(), "wb+");
        if (!f) fatal("FATAL  : can not create " + p);
        return f;
    };
    // the shared files are rank 0's (every rank holds the same chain); the others write theirs into the void
    auto open_shared = [&](const std::string& p) { return open_trunc(rank == 0 ? p : std::string("/dev/null")); };
    FILE* outf = open_shared(base + ".csv");
    FILE* betf = open_shared(base + ".bet");
    FILE* cpnf = open_shared(base + ".cpn");
    FILE* xbetf = open_shared(base + ".xbet");
    FILE* xcpnf = open_shared(base + ".xcpn");
    FILE* gamf = open_shared(base + ".gam"); // text, one line per thinned iteration (:1966-1977)
    FILE* xivf = open_shared(base + ".xiv");
    FILE* epsf = open_trunc(base + ".eps." + std::to_string(rank));
    FILE* mrkf = open_trunc(base + ".mrk." + std::to_string(rank));
    for (FILE* f : {betf, xbetf, cpnf, xcpnf}) pwrite_at(f, 0, &Mtot, sizeof(unsigned)); // :1096-1101

    std::vector<double> beta(Mtot), eps(hi - lo), sigmaG(G), gamma(C);
    std::vector<int32_t> comp(Mtot), m0(G), xiv(C);
    std::vector<char> buff(50000);
    unsigned n_thinned_saved = 0;
    const double t_all = now_s();
    for (unsigned iteration = iteration_start; iteration < opt.chainLength; ++iteration) {
        const double t0 = now_s();
        hg_check(hydraw_chain_iterate(chain), "hydraw_chain_iterate");
        const double t1 = now_s();
        double mu = 0, alpha = 0;
        hydraw_chain_state(chain, &mu, &alpha, sigmaG.data(), nullptr, m0.data(), nullptr, nullptr, nullptr);
        double sg = 0;
        long m0s = 0;
        for (int g = 0; g < G; ++g) {
            sg += sigmaG[g];
            m0s += m0[g];
        }
        if (rank == 0) std::printf("%u. %ld; %.7g; %.7g; %.7g\n", iteration, m0s, mu, alpha, sg); // :1906-1908
        if (rank == 0) std::printf("RESULT : it %4d, rank %4d: proc = %9.3f s, sync = %9.3f (%9.3f + %9.3f), n_sync = %8d (%8d + %8d) (%7.3f / %7.3f), "
                    "betasq = %15.10f, m0 = %10d\n",
                    iteration, rank, t1 - t0, 0.0, 0.0, 0.0, 0, 0, 0, 0.0, 0.0, 0.0, (int)m0s);
        std::fflush(stdout);

        if (iteration % opt.thin == 0) { // :1937-2019
            const int len = hydraw_chain_csv_line(chain, iteration, buff.data(), buff.size());
            pwrite_at(outf, (long)n_thinned_saved * len, buff.data(), (size_t)len);
            if (opt.covariates) {
                hydraw_chain_gamma(chain, gamma.data(), xiv.data());
                std::string gl(64 + 32 * (size_t)C, '\0');
                int n = std::snprintf(&gl[0], gl.size(), "%5d", iteration);
                for (int ii = 0; ii < C; ++ii) n += std::snprintf(&gl[n], gl.size() - n, ", %20.17f", gamma[ii]);
                n += std::snprintf(&gl[n], gl.size() - n, "\n");
                pwrite_at(gamf, (long)n_thinned_saved * n, gl.data(), (size_t)n);
                if (iteration > 0 && iteration % opt.save == 0) {
                    const unsigned nf = (unsigned)C;
                    pwrite_at(xivf, 0, &iteration, sizeof(unsigned));
                    pwrite_at(xivf, sizeof(unsigned), &nf, sizeof(unsigned));
                    pwrite_at(xivf, 2 * sizeof(unsigned), xiv.data(), (size_t)C * sizeof(int));
                }
            }
            hg_check(hgibbs_w_get_beta(dev, beta.data(), comp.data()), "hgibbs_w_get_beta");
            long off = sizeof(unsigned) + (long)n_thinned_saved * (sizeof(unsigned) + (long)Mtot * sizeof(double));
            pwrite_at(betf, off, &iteration, sizeof(unsigned));
            pwrite_at(betf, off + sizeof(unsigned), beta.data(), (size_t)Mtot * sizeof(double));
            off = sizeof(unsigned) + (long)n_thinned_saved * (sizeof(unsigned) + (long)Mtot * sizeof(int));
            pwrite_at(cpnf, off, &iteration, sizeof(unsigned));
            pwrite_at(cpnf, off + sizeof(unsigned), comp.data(), (size_t)Mtot * sizeof(int));
            n_thinned_saved += 1;
        }
        if (iteration > 0 && iteration % opt.save == 0) { // :2028-2052
            hg_check(hydraw_chain_reseed_ars(chain, opt.seed + iteration), "hydraw_chain_reseed_ars"); // srand(opt.seed + iteration)
            hgibbs_rng_state rst;
            hydraw_chain_state(chain, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &rst, nullptr);
            write_rng_file(base + ".rng." + std::to_string(rank), rst);
            hg_check(hgibbs_get_residual(dev, eps.data()), "hgibbs_get_residual");
            hg_check(hgibbs_w_get_beta(dev, beta.data(), comp.data()), "hgibbs_w_get_beta");
            pwrite_at(epsf, 0, &iteration, sizeof(unsigned));
            const unsigned nloc = hi - lo; // this rank's rows
            pwrite_at(epsf, sizeof(unsigned), &nloc, sizeof(unsigned));
            pwrite_at(epsf, 2 * sizeof(unsigned), eps.data(), (size_t)nloc * sizeof(double));
            pwrite_at(mrkf, 0, &iteration, sizeof(unsigned));
            pwrite_at(mrkf, sizeof(unsigned), &Mtot, sizeof(unsigned));
            pwrite_at(mrkf, 2 * sizeof(unsigned), hydraw_chain_order(chain), (size_t)Mtot * sizeof(int));
            pwrite_at(xbetf, sizeof(unsigned), &iteration, sizeof(unsigned));
            pwrite_at(xcpnf, sizeof(unsigned), &iteration, sizeof(unsigned));
            pwrite_at(xbetf, 2 * sizeof(unsigned), beta.data(), (size_t)Mtot * sizeof(double));
            pwrite_at(xcpnf, 2 * sizeof(unsigned), comp.data(), (size_t)Mtot * sizeof(int));
        }
    }
    std::printf("INFO   : rank %4d, time to process the data: %.3f sec\n", rank, now_s() - t_all);
    for (FILE* f : {outf, betf, cpnf, xbetf, xcpnf, gamf, xivf, epsf, mrkf})
        if (f) std::fclose(f);
    hydraw_chain_destroy(chain);
    hgibbs_destroy(dev);
    return 0;
}

} // namespace

int main(int argc, const char* argv[])
{
    if (argc < 2) {
        std::cerr << " \nDid you forget to give the input parameters?\n" << std::endl;
        return 1;
    }
    Options opt = parse(argc, argv);
    if (!((opt.bayesType == "bayesMPI" || opt.bayesType == "bayesWMPI") && opt.analysisType == "RAM")) {
        std::cerr << "\n Error: Wrong analysis requested: " << opt.analysisType << " + " << opt.bayesType
                  << " (this build reproduces --mpibayes bayesMPI and bayesWMPI)" << std::endl;
        return 0; // the reference catches the throw and still returns 0 (main.cpp:179-189)
    }
    if (!opt.readFromBedFile) fatal("FATAL: either go for BED, SPARSE or BOTH (this build reads --bfile)");
    if ((opt.groupIndexFile.empty()) != (opt.groupMixtureFile.empty()))
        fatal("FATAL   : you need to activate both --groupIndexFile and --groupMixtureFile");
    if (opt.syncRate > 1)
        std::printf("WARNING: --sync-rate %d ignored: individuals are sharded, every marker sees the current residual\n", opt.syncRate);

    const char* e;
    const int rank = (e = std::getenv("RANK")) ? std::atoi(e) : 0;
    const int nranks = (e = std::getenv("WORLD_SIZE")) ? std::atoi(e) : 1;
    const int local_rank = (e = std::getenv("LOCAL_RANK")) ? std::atoi(e) : rank;
    if (opt.bayesType == "bayesWMPI") return run_bayesw(opt, rank, nranks, local_rank); // main.cpp:164-167

    // ---- inputs (main.cpp:69-70,88; BayesRRm.cpp:969-997) -------------------
    std::vector<std::string> fam_ids;
    const size_t numInds = count_fam(opt.bedFile + ".fam", &fam_ids);
    const size_t numSnps = count_bim(opt.bedFile + ".bim");
    std::vector<double> y;
    std::vector<uint8_t> keep;
    std::vector<double> covX;
    int C = 0;
    if (opt.covariates) read_phen_cov(opt.phenotypeFile, opt.covariatesFile, numInds, y, keep, covX, C); // main.cpp:80-83
    else read_phen(opt.phenotypeFile, fam_ids, y, keep);
    const unsigned numNAs = (unsigned)(numInds - y.size());

    if (opt.numberIndividuals == 0) fatal("FATAL  : opt.numberIndividuals is zero! Set it via --number-individuals in call.");
    if (opt.numberMarkers == 0) fatal("FATAL  : opt.numberMarkers is zero! Set it via --number-markers in call.");
    if (opt.numberIndividuals != numInds) fatal("FATAL  : --number-individuals does not match the .fam file");
    unsigned Mtot = opt.numberMarkers;
    if (Mtot > numSnps) fatal("FATAL  : --number-markers exceeds the .bim file");
    if (Mtot < numSnps && rank == 0) std::printf("INFO   : Option passed to process only %d markers!\n", Mtot);
    const unsigned Ntot = (unsigned)numInds - numNAs;
    if (rank == 0) {
        if (numNAs)
            std::printf("WARNING: opt.numberIndividuals set to %zu but will be adjusted to %zu - %u = %u due to NAs in phenotype file.\n",
                        numInds, numInds, numNAs, Ntot);
        std::printf("INFO   : Full dataset includes Mtot=%d markers and Ntot=%d individuals.\n", Mtot, (int)numInds);
    }

    std::vector<int32_t> groups;
    std::vector<std::vector<double>> mS;
    if (!opt.groupIndexFile.empty()) {
        groups = read_groups(opt.groupIndexFile);
        mS = read_mS(opt.groupMixtureFile);
        if (groups.size() < Mtot) fatal("FATAL  : group file covers fewer markers than --number-markers");
        groups.resize(Mtot);
    } else {
        std::vector<double> row{0.0};
        for (double v : opt.S) {
            if (v <= 0.0) fatal("FATAL  : mixture value can only be strictly positive");
            row.push_back(v);
        }
        mS.push_back(row);
    }
    const int G = (int)mS.size(), K = (int)mS[0].size();
    std::vector<double> mS_flat;
    for (auto& r : mS) mS_flat.insert(mS_flat.end(), r.begin(), r.end());

    // --thin/--save adjustment, BayesRRm.cpp:1058-1066
    if (opt.save < opt.thin) {
        opt.save = opt.thin;
        if (rank == 0) std::printf("WARNING: opt.save was lower that opt.thin ; opt.save reset to opt.thin (%d)\n", opt.thin);
    }
    if (opt.save % opt.thin != 0) {
        opt.save = (opt.save / opt.thin) * opt.thin;
        if (rank == 0) std::printf("         opt.save reset to %d, the closest multiple of opt.thin (%d)\n", opt.save, opt.thin);
    }

    struct stat sb;
    if (stat(opt.mcmcOutDir.c_str(), &sb) != 0 && rank == 0)
        if (std::system(("mkdir -p " + opt.mcmcOutDir).c_str()) != 0) fatal("FATAL  : can not create --mcmc-out-dir");
    // a restart reads <name>.* and writes <name>_rs.* so the failed job's files stay untouched (:1206-1220)
    const std::string base_in = opt.mcmcOutDir + "/" + opt.mcmcOutNam;
    const std::string base = opt.restart ? base_in + "_rs" : base_in;

    // ---- device -------------------------------------------------------------
    hgibbs_t dev = nullptr;
    hg_check(hgibbs_create(local_rank, &dev), "hgibbs_create");
    if (nranks > 1) setup_ranks(dev, base, rank, nranks);
    if (opt.batch) hg_check(hgibbs_set_option(dev, "batch", opt.batch), "batch");
    if (opt.cpg) hg_check(hgibbs_set_option(dev, "cols_per_group", opt.cpg), "cols_per_group");

    // ---- genotypes: Data::load_data_from_bed_file, data.cpp:671-739 -----------
    const double tl0 = now_s();
    const size_t snpLenByt = (numInds + 3) / 4;
    std::vector<uint8_t> bed;
    {
        std::ifstream in(opt.bedFile + ".bed", std::ios::binary);
        if (!in) fatal("Error: can not open the file [" + opt.bedFile + ".bed] to read.");
        unsigned char magic[3];
        in.read((char*)magic, 3);
        if (!in || magic[0] != 0x6c || magic[1] != 0x1b || magic[2] != 0x01) fatal("FATAL  : " + opt.bedFile + ".bed is not a SNP-major PLINK bed");
        bed.resize((size_t)Mtot * snpLenByt);
        in.read((char*)bed.data(), (std::streamsize)bed.size());
        if ((size_t)in.gcount() != bed.size()) fatal("FATAL  : " + opt.bedFile + ".bed is shorter than M x ceil(N/4)");
    }
    // individuals sharded in multiples of 4 of the KEPT rows
    const unsigned per = ((Ntot + nranks - 1) / nranks + 3) / 4 * 4;
    const unsigned lo = std::min(Ntot, rank * per), hi = std::min(Ntot, (rank + 1) * per);
    hg_check(hgibbs_load_bed(dev, bed.data(), snpLenByt, (uint32_t)numInds, Mtot, numNAs ? keep.data() : nullptr, lo, hi, Ntot),
             "hgibbs_load_bed");
    std::printf("INFO   : rank %3d took %.3f seconds to load  %lu bytes  =>  BW = %7.3f GB/s\n", rank, now_s() - tl0,
                (unsigned long)bed.size(), (double)bed.size() * 1e-9 / (now_s() - tl0));
    std::vector<uint8_t>().swap(bed);

    hydra_model_desc md{};
    md.seed = opt.seed + (unsigned)0 * 1000; // every rank replicates rank 0's stream (BayesRRm.cpp:1228 with rank = 0)
    md.shuffle = opt.shuffleMarkers;
    md.G = G;
    md.K = K;
    md.groups = groups.empty() ? nullptr : groups.data();
    md.mS = mS_flat.data();
    hydra_chain_t chain = nullptr;
    hg_check(hydra_chain_create(dev, &md, y.data(), &chain), "hydra_chain_create");
    if (opt.covariates) {
        if (rank == 0) std::printf("INFO   : using covariate file: %s\n", opt.covariatesFile.c_str());
        hg_check(hydra_chain_set_covariates(chain, covX.data(), C), "hydra_chain_set_covariates");
    }

    // ---- --restart: BayesRRm::init_from_restart, :842-928, then :1546-1597 -----
    unsigned iteration_start = 0;
    if (opt.restart) {
        if (rank == 0) std::printf("RESTART: from files: %s.* files\n", base_in.c_str());
        const CsvRestart cr = read_csv_for_restart(base_in + ".csv", opt.thin, opt.save, G, K);
        if (rank == 0) {
            std::printf("RESTART: Reading .cvs file %s\n", (base_in + ".csv").c_str());
            std::printf("RESTART: --thin %d  -- save %d\n", opt.thin, opt.save);
            std::printf("RESTART: iteration_to_restart_from = %d\n", cr.iteration_to_restart_from);
            std::printf("RESTART: first_thinned_iteration   = %d\n", cr.first_thinned_iteration);
            std::printf("RESTART: first_saved_iteration     = %d\n", cr.first_saved_iteration);
        }
        if (cr.iteration_to_restart_from == 0)
            fatal("FATAL  : There is no point in restarting a chain from iteration 0 (not saved anyway)\n         => restart your analysis from scratch");
        const unsigned it_from = cr.iteration_to_restart_from;
        const bool xf = opt.useXfilesInRestart;
        std::vector<double> r_beta(Mtot);
        std::vector<int32_t> r_comp(Mtot);
        read_marker_history(base_in + (xf ? ".xbet" : ".bet"), xf, Mtot, it_from, cr.first_thinned_iteration, opt.thin, sizeof(double), r_beta.data());
        read_marker_history(base_in + (xf ? ".xcpn" : ".cpn"), xf, Mtot, it_from, cr.first_thinned_iteration, opt.thin, sizeof(int32_t), r_comp.data());
        double r_mu = 0.0;
        { // .mus.<rank>: (iteration, mu) per thinned iteration, data.cpp:207-252
            const std::string mp = base_in + ".mus." + std::to_string(rank);
            FILE* f = open_ro(mp);
            const long off = (long)((it_from - cr.first_thinned_iteration) / opt.thin) * (long)(sizeof(unsigned) + sizeof(double));
            unsigned it_ = ~0u;
            pread_at(f, off, &it_, sizeof(unsigned), mp);
            expect_u(it_, it_from, mp + " iteration");
            pread_at(f, off + (long)sizeof(unsigned), &r_mu, sizeof(double), mp);
            std::fclose(f);
        }
        unsigned len = 0;
        // .eps.<rank>: this rank's shard; a full-length dump (hydra's own, or a 1-rank run's) is sliced
        std::vector<uint8_t> eb = read_dump(base_in + ".eps." + std::to_string(rank), it_from, sizeof(double), &len);
        const double* r_eps = (const double*)eb.data();
        if (len == Ntot && len != hi - lo) r_eps += lo;
        else expect_u(len, hi - lo, ".eps Ntot");
        std::vector<uint8_t> mb = read_dump(base_in + ".mrk." + std::to_string(rank), it_from, sizeof(int32_t), &len);
        expect_u(len, Mtot, ".mrk M");
        std::vector<uint8_t> gb, xb;
        if (opt.covariates) {
            gb = read_dump(base_in + ".gam." + std::to_string(rank), it_from, sizeof(double), &len);
            expect_u(len, (unsigned)C, ".gam length");
            xb = read_dump(base_in + ".xiv." + std::to_string(rank), it_from, sizeof(int32_t), &len);
            expect_u(len, (unsigned)C, ".xiv length");
        }
        hydra_restart_state rs{};
        rs.iteration = it_from;
        rs.sigmaE = cr.sigmaE;
        rs.mu = r_mu;
        rs.sigmaG = cr.sigmaG.data();
        rs.estPi = cr.pi.data();
        rs.beta = r_beta.data();
        rs.components = r_comp.data();
        rs.eps = r_eps;
        rs.order = (const int32_t*)mb.data();
        rs.gamma = opt.covariates ? (const double*)gb.data() : nullptr;
        rs.xI = opt.covariates ? (const int32_t*)xb.data() : nullptr;
        read_rng_file(base_in + ".rng." + std::to_string(rank), &rs.rng);
        hg_check(hydra_chain_restore(chain, &rs), "hydra_chain_restore");
        iteration_start = it_from + 1;
        if (rank == 0) { // Data::print_restart_banner, data.cpp:20-31
            std::printf("INFO   : %s\n", std::string(100, '*').c_str());
            std::printf("INFO   : RESTART DETECTED\n");
            std::printf("INFO   : restarting from: %s.* files\n", base_in.c_str());
            std::printf("INFO   : last saved iteration:        %d\n", it_from);
            std::printf("INFO   : will restart from iteration: %d\n", iteration_start);
            std::printf("INFO   : %s\n", std::string(100, '*').c_str());
        }
    }

    // ---- outputs (rank 0 writes the shared files) ----------------------------
    FILE *outf = nullptr, *betf = nullptr, *cpnf = nullptr, *acuf = nullptr, *xbetf = nullptr, *xcpnf = nullptr;
    auto open_trunc = [&](const std::string& p) {
        FILE* f = std::fopen(p.c_str(), "wb+");
        if (!f) fatal("FATAL  : can not create " + p);
        return f;
    };
    if (rank == 0) {
        outf = open_trunc(base + ".csv");
        betf = open_trunc(base + ".bet");
        cpnf = open_trunc(base + ".cpn");
        acuf = open_trunc(base + ".acu");
        xbetf = open_trunc(base + ".xbet");
        xcpnf = open_trunc(base + ".xcpn");
        for (FILE* f : {betf, xbetf, cpnf, xcpnf, acuf}) pwrite_at(f, 0, &Mtot, sizeof(unsigned)); // :1302-1309
    }
    FILE* musf = open_trunc(base + ".mus." + std::to_string(rank));
    FILE* epsf = open_trunc(base + ".eps." + std::to_string(rank));
    FILE* mrkf = open_trunc(base + ".mrk." + std::to_string(rank));
    FILE* gamf = open_trunc(base + ".gam." + std::to_string(rank));
    FILE* xivf = open_trunc(base + ".xiv." + std::to_string(rank));
    std::vector<double> gamma(C);
    std::vector<int32_t> xiv(C);

    std::vector<double> beta(Mtot), acum(Mtot), eps(hi - lo);
    std::vector<int32_t> comp(Mtot);
    std::vector<double> sigmaG(G);
    std::vector<int32_t> m0(G);
    std::vector<char> buff(50000);
    unsigned n_thinned_saved = 0;
    const double t_all = now_s();

    for (unsigned iteration = iteration_start; iteration < opt.chainLength; ++iteration) {
        const double t0 = now_s();
        hg_check(hydra_chain_iterate(chain), "hydra_chain_iterate");
        const double t1 = now_s();
        double sigmaE = 0, mu = 0;
        hydra_chain_state(chain, &sigmaE, &mu, sigmaG.data(), nullptr, m0.data(), nullptr, nullptr);
        double sg = 0;
        long m0s = 0;
        for (int g = 0; g < G; ++g) {
            sg += sigmaG[g];
            m0s += m0[g];
        }
        if (rank % 10 == 0) { // :2713-2722 (sync columns are zero: there is no marker-sharded sync in this build)
            std::printf("RESULT : it %4d, rank %4d: proc = %9.3f s, sync = %9.3f (%9.3f + %9.3f), n_sync = %8d (%8d + %8d) (%7.3f / %7.3f), "
                        "sigmaG = %15.10f, sigmaE = %15.10f, betasq = %15.10f, m0 = %10ld\n",
                        iteration, rank, t1 - t0, 0.0, 0.0, 0.0, 0, 0, 0, 0.0, 0.0, sg, sigmaE, 0.0, m0s);
            std::fflush(stdout);
        }

        if (iteration % opt.thin == 0) { // :2736-2794
            hg_check(hgibbs_get_beta(dev, beta.data(), comp.data(), acum.data()), "hgibbs_get_beta");
            if (rank == 0) {
                const int len = hydra_chain_csv_line(chain, iteration, buff.data(), buff.size());
                pwrite_at(outf, (long)n_thinned_saved * len, buff.data(), (size_t)len);
                long off = sizeof(unsigned) + (long)n_thinned_saved * (sizeof(unsigned) + (long)Mtot * sizeof(double));
                pwrite_at(betf, off, &iteration, sizeof(unsigned));
                pwrite_at(acuf, off, &iteration, sizeof(unsigned));
                pwrite_at(betf, off + sizeof(unsigned), beta.data(), (size_t)Mtot * sizeof(double));
                pwrite_at(acuf, off + sizeof(unsigned), acum.data(), (size_t)Mtot * sizeof(double));
                off = sizeof(unsigned) + (long)n_thinned_saved * (sizeof(unsigned) + (long)Mtot * sizeof(int));
                pwrite_at(cpnf, off, &iteration, sizeof(unsigned));
                pwrite_at(cpnf, off + sizeof(unsigned), comp.data(), (size_t)Mtot * sizeof(int));
            }
            long off = (long)n_thinned_saved * (sizeof(unsigned) + sizeof(double));
            pwrite_at(musf, off, &iteration, sizeof(unsigned));
            pwrite_at(musf, off + sizeof(unsigned), &mu, sizeof(double));
            n_thinned_saved += 1;
        }

        if (iteration > 0 && iteration % opt.save == 0) { // :2802-2838 (no tarball)
            hgibbs_rng_state rst;
            hydra_chain_state(chain, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &rst);
            write_rng_file(base + ".rng." + std::to_string(rank), rst);
            hg_check(hgibbs_get_residual(dev, eps.data()), "hgibbs_get_residual");
            const unsigned nloc = hi - lo;
            pwrite_at(epsf, 0, &iteration, sizeof(unsigned));
            pwrite_at(epsf, sizeof(unsigned), &nloc, sizeof(unsigned));
            pwrite_at(epsf, 2 * sizeof(unsigned), eps.data(), (size_t)nloc * sizeof(double));
            pwrite_at(mrkf, 0, &iteration, sizeof(unsigned));
            pwrite_at(mrkf, sizeof(unsigned), &Mtot, sizeof(unsigned));
            pwrite_at(mrkf, 2 * sizeof(unsigned), hydra_chain_order(chain), (size_t)Mtot * sizeof(int));
            if (opt.covariates) { // :2810-2832
                const unsigned glen = (unsigned)C;
                hydra_chain_gamma(chain, gamma.data(), xiv.data());
                for (FILE* f : {gamf, xivf}) {
                    pwrite_at(f, 0, &iteration, sizeof(unsigned));
                    pwrite_at(f, sizeof(unsigned), &glen, sizeof(unsigned));
                }
                pwrite_at(gamf, 2 * sizeof(unsigned), gamma.data(), (size_t)C * sizeof(double));
                pwrite_at(xivf, 2 * sizeof(unsigned), xiv.data(), (size_t)C * sizeof(int));
            }
            if (rank == 0) {
                hg_check(hgibbs_get_beta(dev, beta.data(), comp.data(), nullptr), "hgibbs_get_beta");
                pwrite_at(xbetf, sizeof(unsigned), &iteration, sizeof(unsigned));
                pwrite_at(xcpnf, sizeof(unsigned), &iteration, sizeof(unsigned));
                pwrite_at(xbetf, 2 * sizeof(unsigned), beta.data(), (size_t)Mtot * sizeof(double));
                pwrite_at(xcpnf, 2 * sizeof(unsigned), comp.data(), (size_t)Mtot * sizeof(int));
            }
        }
    }
    if (rank == 0)
        std::printf("INFO   : rank %4d, time to process the data: %.3f sec, with %.3f (%.3f, %.3f) = %4.1f%% spent on allred (%d, %d)\n", rank,
                    now_s() - t_all, 0.0, 0.0, 0.0, 0.0, 0, 0);
    for (FILE* f : {outf, betf, cpnf, acuf, xbetf, xcpnf, musf, epsf, mrkf, gamf, xivf})
        if (f) std::fclose(f);
    hydra_chain_destroy(chain);
    hgibbs_destroy(dev);
    return 0;
}
