/*
 * oracle/shuffle_helper.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The random serial schedule of the reference (bp.hpp:467-468) is `rng_list_shuffle.shuffle(serial_schedule_order)` once per
 * iteration: std::shuffle on a std::mt19937 seeded by random_schedule_seed (rng.hpp:84-133).  Both are the C++ standard
 * library's (libstdc++: third-party to the reference); this helper calls exactly those, so the orders it returns are the
 * ones a reference decoder object walks through.  State in / state out, so that a sequence of decodes on ONE decoder object
 * (whose generator and order live on from decode to decode) can be followed as well.
 */
#include <algorithm>
#include <cstdint>
#include <random>
#include <sstream>
#include <string>
#include <vector>

extern "C" {

/* orders_out [count][n]: the arrangement of `order` (n entries, updated in place) after 1, 2, ..., count shuffles.
 * state: textual std::mt19937 state, in and out (buffer of >= 8192 bytes); an empty string = freshly seeded with `seed`. */
void oracle_shuffle_orders(uint32_t seed, int32_t n, int32_t *order, int32_t count, int32_t *orders_out, char *state) {
    std::mt19937 g;
    if (state && state[0]) { std::istringstream is(state); is >> g; } else g.seed(seed);
    std::vector<int> v(order, order + n);
    for (int c = 0; c < count; ++c) {
        std::shuffle(v.begin(), v.end(), g);
        if (orders_out) std::copy(v.begin(), v.end(), orders_out + (size_t)c * (size_t)n);
    }
    std::copy(v.begin(), v.end(), order);
    if (state) { std::ostringstream os; os << g; const std::string s = os.str(); s.copy(state, s.size()); state[s.size()] = 0; }
}

/* soft_info_decode_serial's random schedule (bp.hpp:573-577) is a different draw: `shuffle(order, std::default_random_engine(
 * random_schedule_seed))` -- a NEW engine from the same seed at the top of every iteration, i.e. one fixed rearrangement applied
 * again and again to the order the object carries.  orders_out [count][n] as above; `order` is updated in place. */
void oracle_shuffle_orders_reseeded(int32_t seed, int32_t n, int32_t *order, int32_t count, int32_t *orders_out) {
    std::vector<int> v(order, order + n);
    for (int c = 0; c < count; ++c) {
        std::shuffle(v.begin(), v.end(), std::default_random_engine(seed));
        if (orders_out) std::copy(v.begin(), v.end(), orders_out + (size_t)c * (size_t)n);
    }
    std::copy(v.begin(), v.end(), order);
}

}
