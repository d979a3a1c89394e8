// dphpc_amd: driver with the command line of the reference's `dphpc` binary (src/main.cpp:23-28,
// 47-93), running the multiply on MI355X GPUs through the C ABI.
//
//   dphpc_amd <algo> <matrix> <run_path> <n_runs> <n_warmup> <shuffle> <partition>
//             <parallel_load> <persist> <matrix_path>
//             [--ranks P] [--gpus G] [--impl baseline|drop|shuffle] [--sections N] [--seed S] [--shuffle-iterations I] [--relabel result|b]
//             [--stages S]
//
// <algo>: baseline | outer (no section dropping) | drop | drop_parallel | drop_at_once |
//         drop_at_once_parallel (bitmap-filtered exchange).  All run the same GPU operator; the
//         name only selects whether the bitmap is used.  `full` and `comb` are not built.
// --impl overrides <algo>/<shuffle>: baseline = no drop, no shuffle; drop = bitmap; shuffle =
//         bitmap + random row/column shuffle.
// --stages S: the pipelined schedule: every rank cuts its rows into S blocks and exchanges the B sections
//         of block s + 1 while block s is multiplied (HipOptions::stages).
// The P ranks live in this one process (rank r on GPU r % G); the reference starts P MPI ranks.
// Files written to <run_path>: algo, matrix, partitions.csv, A_shuffle, B_shuffle, C_<rank>.mtx,
// measurements_<rank>.csv: the layout the reference's scripts read.
// B is <matrix_path>/<matrix>/B.mtx when that file exists, else A.mtx (the reference's HEAD
// multiplies A by itself, src/main.cpp:64-70; its `test` fixture expects A*B).
#include "mults.hpp"

#include <cstdio>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <memory>
#include <thread>

static void write_to_file(const std::string &name, const std::string &path) {
  std::ofstream f(path);
  f << name << std::endl;
}

int main(int argc, char **argv) {
  try {
    std::vector<std::string> pos;
    int n_ranks = 0, n_gpus = 0;
    uint32_t n_sections = 512, n_stages = 1;
    uint64_t seed = 1;
    std::string relabel_where;         // "", "result" or "b": where a column shuffle is applied
    uint32_t shuffle_iterations = 20; // of the iterative shuffle (the reference stops on wall clock)
    std::string impl;
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      auto next = [&]() -> std::string {
        if (i + 1 >= argc) { std::cerr << "missing value after " << a << "\n"; exit(1); }
        return argv[++i];
      };
      if (a == "--ranks") n_ranks = std::stoi(next());
      else if (a == "--gpus") n_gpus = std::stoi(next());
      else if (a == "--impl") impl = next();
      else if (a == "--sections") n_sections = (uint32_t)std::stoul(next());
      else if (a == "--seed") seed = std::stoull(next());
      else if (a == "--shuffle-iterations") shuffle_iterations = (uint32_t)std::stoul(next());
      else if (a == "--relabel") relabel_where = next();
      else if (a == "--stages") n_stages = (uint32_t)std::stoul(next());
      else pos.push_back(a);
    }
    if (pos.size() < 10) {
      std::cerr << "Did not get enough arguments. Expected <algo> <matrix> <run_path> <n_runs> <n_warmup> <shuffle> "
                   "<partition> <parallel_load> <persist> <matrix_path>" << std::endl;
      return 1;
    }
    std::string algo_name = pos[0], matrix_name = pos[1], run_path = pos[2];
    int n_runs = std::stoi(pos[3]), n_warmup = std::stoi(pos[4]);
    std::string shuffling_algo = pos[5], partitioning_algo = pos[6];
    bool persist_results = pos[8] != "false";
    std::string matrix_path = pos[9];
    if (!persist_results) std::cout << "CAUTION: NOT PERSISTING RESULTS" << std::endl;

    bool drop;
    if (algo_name == "baseline" || algo_name == "outer") drop = false;
    else if (algo_name == "drop" || algo_name == "drop_parallel" || algo_name == "drop_at_once" ||
             algo_name == "drop_at_once_parallel") drop = true;
    else { std::cerr << "Unknown algorithm type " << algo_name << "\n"; return 1; }
    if (!impl.empty()) {
      if (impl == "baseline") { drop = false; shuffling_algo = "none"; }
      else if (impl == "drop") { drop = true; }
      else if (impl == "shuffle") { drop = true; shuffling_algo = "random"; }
      else { std::cerr << "Unknown --impl " << impl << "\n"; return 1; }
    }

    int dev_count = 0;
    if (spgemm_device_count(&dev_count) != SPGEMM_OK || dev_count < 1) {
      std::cerr << "no HIP device: " << spgemm_last_error() << std::endl;
      return 2;
    }
    if (n_gpus <= 0 || n_gpus > dev_count) n_gpus = dev_count;
    if (n_ranks <= 0) n_ranks = n_gpus;
    const int n_nodes = n_ranks;

    std::string A_path = matrix_path + "/" + matrix_name + "/A.mtx";
    std::string B_path = matrix_path + "/" + matrix_name + "/B.mtx";
    if (!std::filesystem::exists(B_path)) B_path = A_path;
    std::filesystem::create_directories(run_path);
    measure_point(0, measure::global, measure::MeasurementEvent::START);
    write_to_file(matrix_name, run_path + "/matrix");
    write_to_file(algo_name, run_path + "/algo");

    measure_point(0, measure::read_triplets, measure::MeasurementEvent::START);
    auto tA = matrix::utils::read_matrix_market<double>(A_path);
    matrix::Triplets<double> tB_store;
    const matrix::Triplets<double> *tB = &tA;
    if (B_path != A_path) { tB_store = matrix::utils::read_matrix_market<double>(B_path); tB = &tB_store; }
    measure_point(0, measure::read_triplets, measure::MeasurementEvent::END);
    if (tA.ncols != tB->nrows) { std::cerr << "A.cols != B.rows\n"; return 1; }

    // shuffles (reference src/main.cpp:129-161)
    partition::Shuffle A_shuffle(tA.nrows), B_shuffle(tB->ncols);
    std::cout << "STARTING SHUFFLING!" << std::endl;
    std::string A_sh_saved = matrix_path + "/" + matrix_name + "/A_shuffle_" + shuffling_algo;
    std::string B_sh_saved = matrix_path + "/" + matrix_name + "/B_shuffle_" + shuffling_algo;
    measure_point(0, measure::shuffle, measure::MeasurementEvent::START);
    if (!(partition::load_shuffle(A_sh_saved, A_shuffle) && partition::load_shuffle(B_sh_saved, B_shuffle))) {
      if (shuffling_algo == "none") { A_shuffle = partition::identity(tA.nrows); B_shuffle = partition::identity(tB->ncols); }
      else if (shuffling_algo == "random") { A_shuffle = partition::shuffle(tA.nrows, seed); B_shuffle = partition::shuffle(tB->ncols, seed + 1); }
      else if (shuffling_algo == "iterative") {
        matrix::ManagedCSRMatrix<> hA(tA, false, nullptr, nullptr), hB(*tB, false, nullptr, nullptr);
        partition::iterative_shuffle(hA, hB, 0, shuffle_iterations, &A_shuffle, &B_shuffle);
      }
      else { std::cerr << "Unknown shuffling algorithm type " << shuffling_algo << "\n"; return 1; }
    }
    measure_point(0, measure::shuffle, measure::MeasurementEvent::END);
    partition::save_shuffle(A_shuffle, run_path + "/A_shuffle");
    partition::save_shuffle(B_shuffle, run_path + "/B_shuffle");
    std::cout << "SHUFFLING FINISHED!" << std::endl;

    if (n_stages < 1 || n_stages > 64) { std::cerr << "--stages takes 1..64\n"; return 1; }
    communication::World world(n_nodes, n_gpus, n_stages > 1);

    // partitions (reference src/main.cpp:176-199)
    std::cout << "Computing \"" << partitioning_algo << "\" partitioning" << std::endl;
    measure_point(0, measure::partition, measure::MeasurementEvent::START);
    partition::Partitions partitions(n_nodes);
    matrix::Fields cf{false, tA.nrows, tB->ncols, 0};
    if (partitioning_algo == "naive") {
      partitions = parts::baseline::partition(cf, n_nodes);
    } else if (partitioning_algo == "balanced") {
      // weight of a (shuffled) row = its number of products; needs no C_sparsity file
      std::vector<midx_t> all_rows(A_shuffle.begin(), A_shuffle.end());
      matrix::ManagedCSRMatrix<> hA(tA, false, &all_rows, nullptr), hB(*tB, false, nullptr, nullptr);
      std::vector<uint64_t> len_b(hB.height);
      for (midx_t r = 0; r < hB.height; ++r) len_b[r] = hB.row_ptr[r + 1] - hB.row_ptr[r];
      std::vector<uint64_t> w(hA.height, 0);
      for (midx_t r = 0; r < hA.height; ++r)
        for (midx_t j = hA.row_ptr[r]; j < hA.row_ptr[r + 1]; ++j) w[r] += len_b[hA.col_idx[j]];
      partitions = parts::baseline::balanced_partition(w, tB->ncols, n_nodes);
    } else { std::cerr << "Unknown partitioning algorithm type " << partitioning_algo << "\n"; return 1; }
    measure_point(0, measure::partition, measure::MeasurementEvent::END);
    if (persist_results) partition::save_partitions(partitions, run_path + "/partitions.csv");

    // operators, one per rank (reference src/main.cpp:223-310)
    mults::HipOptions opt{drop, n_sections, mults::product_rows_are_narrow(tA, *tB), n_stages};
    if (relabel_where == "result") opt.relabel_result = true;
    else if (relabel_where == "b") opt.relabel_result = false;
    else if (!relabel_where.empty()) { std::cerr << "--relabel takes result or b\n"; return 1; }
    std::vector<std::unique_ptr<mults::HipSpgemm>> mult(n_nodes);
    for (int rank = 0; rank < n_nodes; ++rank) {
      std::vector<midx_t> keep_rows(A_shuffle.begin() + partitions[rank].start_row, A_shuffle.begin() + partitions[rank].end_row);
      mult[rank] = std::make_unique<mults::HipSpgemm>(rank, n_nodes, partitions, tA, &keep_rows, *tB, &B_shuffle, world, opt);
    }
    // "Allgather" of the bitmaps and "Alltoall" of the filtered sizes
    std::vector<std::vector<size_t>> sizes_from(n_nodes, std::vector<size_t>(n_nodes));
    for (int r = 0; r < n_nodes; ++r) mult[r]->share_bitmaps(mult);
    for (int r = 0; r < n_nodes; ++r) {
      auto s = mult[r]->get_B_serialization_sizes();
      for (int d = 0; d < n_nodes; ++d) sizes_from[d][r] = s[d];
    }

    // One host thread per rank, as the reference's ranks are processes that run at the same time
    // (src/main.cpp:336-356): warm-up runs, then the timed runs; every run starts behind a barrier
    // (the reference's communication::sync_start_time).
    for (int r = 0; r < n_nodes; ++r) { measure::Measure::get_instance(r); mult[r]->prepare(); }
    std::cout << "STARTING COMPUTATION" << std::endl;
    std::vector<std::string> failure(n_nodes);
    auto rank_main = [&](int r) {
      try {
        for (int i = 0; i < n_warmup + n_runs; ++i) {
          if (i == n_warmup) measure::Measure::get_instance(r)->reset_bytes();
          mult[r]->reset();
          world.barrier();
          if (i >= n_warmup) measure_point(r, measure::gemm, measure::MeasurementEvent::START);
          mult[r]->gemm(sizes_from[r], 0);
          if (i >= n_warmup) {
            measure_point(r, measure::gemm, measure::MeasurementEvent::END);
            measure::Measure::get_instance(r)->flush_bytes();
          }
        }
      } catch (const std::exception &e) {
        // The other ranks may be waiting for this one: call the meetings off (they give up at
        // their next one, with an error of their own) instead of ending the process under them.
        failure[r] = e.what();
        world.abandon();
      }
    };
    const auto t_start = std::chrono::steady_clock::now();
    {
      std::vector<std::thread> threads;
      for (int r = 0; r < n_nodes; ++r) threads.emplace_back(rank_main, r);
      for (auto &t : threads) t.join();
    }
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    bool failed = false;
    for (int r = 0; r < n_nodes; ++r)
      if (!failure[r].empty()) {
        std::cerr << "rank " << r << ": " << failure[r] << std::endl;
        failed = true;
      }
    if (failed) {
      mult.clear(); // the operators go before the world they live in
      return 1;
    }
    printf("ranks %d: %d runs (%d warm-up) in %.3f ms wall, %.3f ms per run\n", n_nodes, n_warmup + n_runs, n_warmup, wall_ms,
           wall_ms / std::max(1, n_warmup + n_runs));
    measure_point(0, measure::global, measure::MeasurementEvent::END);

    double nprod = 0, ms = 0;
    for (int r = 0; r < n_nodes; ++r) { nprod += (double)mult[r]->last_stats.nprod; ms = std::max<double>(ms, mult[r]->last_stats.ms_total); }
    printf("ranks %d gpus %d products %.0f slowest-rank multiply %.3f ms -> %.1f GFLOP/s (useful flops)\n", n_nodes, n_gpus,
           nprod, ms, ms > 0 ? 2.0 * nprod / (ms * 1e-3) / 1e9 : 0.0);

    for (int r = 0; r < n_nodes; ++r) {
      if (persist_results) mult[r]->save_result(run_path + "/C_" + std::to_string(r) + ".mtx");
      measure::Measure::get_instance(r)->save(run_path + "/measurements_" + std::to_string(r) + ".csv");
    }
    mult.clear();
    return 0;
  } catch (const std::exception &e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 1;
  }
}
