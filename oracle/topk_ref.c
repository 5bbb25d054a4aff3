/* ORACLE - test infrastructure only (tests/, __graft_entry__.smoke, bench.py cpu_baseline).
 *
 * CPU restatement of the tie-breaking behaviour of `torch.topk(x, k, dim=1)` (largest=True,
 * sorted=True) on CPU tensors, which is what the reference's two selection sites observe:
 *   /root/reference/modeling/fusion_part/SFTS.py:155      (fp32 scores, k = 2 per head)
 *   /root/reference/modeling/fusion_part/Frequency.py:58  (int32 counts, k = 10)
 * The arithmetic lives in PyTorch (third party, torch==2.10.0+rocm7.0 here; the reference pins
 * torch==1.10.0a0, requirements.txt:157), not under /root/reference.  ATen's CPU kernel builds
 * (value, index) pairs and runs libstdc++'s std::partial_sort when k*64 <= n, else
 * std::nth_element(k-1) followed by std::sort of the first k-1.  The
 * libstdc++ algorithms (GCC 11 bits/stl_algo.h:79-97,1635-1650,1819-1988, bits/stl_heap.h:134-150,
 * 223-266,339-362,405-425) are restated below in C for (value,index) pairs.
 * Pinned by tests/test_oracle_topk.py against torch.topk itself (order AND set).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double v; int64_t i; int nan; } pair_t;

/* comp(x, y): "x goes before y" == x is larger (NaN counts as largest). */
static inline int before(const pair_t *x, const pair_t *y) {
    return (x->nan && !y->nan) || (x->v > y->v);
}

static void push_heap_(pair_t *f, long hole, long top, pair_t val) {
    long parent = (hole - 1) / 2;
    while (hole > top && before(&f[parent], &val)) {
        f[hole] = f[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    f[hole] = val;
}

static void adjust_heap_(pair_t *f, long hole, long len, pair_t val) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (before(&f[child], &f[child - 1])) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    push_heap_(f, hole, top, val);
}

static void make_heap_(pair_t *f, long len) {
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) {
        pair_t v = f[parent];
        adjust_heap_(f, parent, len, v);
        if (parent == 0) return;
        parent--;
    }
}

/* __pop_heap(first, last, result): heap is [first, last) */
static void pop_heap_(pair_t *f, long len, pair_t *result) {
    pair_t v = *result;
    *result = f[0];
    adjust_heap_(f, 0, len, v);
}

static void heap_select_(pair_t *f, long middle, long last) {
    make_heap_(f, middle);
    for (long i = middle; i < last; ++i)
        if (before(&f[i], &f[0])) pop_heap_(f, middle, &f[i]);
}

static void sort_heap_(pair_t *f, long len) {
    while (len > 1) {
        --len;
        pop_heap_(f, len, &f[len]);
    }
}

static void swap_(pair_t *a, pair_t *b) { pair_t t = *a; *a = *b; *b = t; }

static void move_median_to_first_(pair_t *r, pair_t *a, pair_t *b, pair_t *c) {
    if (before(a, b)) {
        if (before(b, c)) swap_(r, b);
        else if (before(a, c)) swap_(r, c);
        else swap_(r, a);
    } else if (before(a, c)) swap_(r, a);
    else if (before(b, c)) swap_(r, c);
    else swap_(r, b);
}

static long unguarded_partition_(pair_t *f, long first, long last, long pivot) {
    for (;;) {
        while (before(&f[first], &f[pivot])) ++first;
        --last;
        while (before(&f[pivot], &f[last])) --last;
        if (!(first < last)) return first;
        swap_(&f[first], &f[last]);
        ++first;
    }
}

static long partition_pivot_(pair_t *f, long first, long last) {
    long mid = first + (last - first) / 2;
    move_median_to_first_(&f[first], &f[first + 1], &f[mid], &f[last - 1]);
    return unguarded_partition_(f, first + 1, last, first);
}

static void unguarded_linear_insert_(pair_t *f, long last) {
    pair_t v = f[last];
    long next = last - 1;
    while (before(&v, &f[next])) {
        f[last] = f[next];
        last = next;
        --next;
    }
    f[last] = v;
}

static void insertion_sort_(pair_t *f, long first, long last) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (before(&f[i], &f[first])) {
            pair_t v = f[i];
            memmove(&f[first + 1], &f[first], (size_t)(i - first) * sizeof(pair_t));
            f[first] = v;
        } else {
            unguarded_linear_insert_(f, i);
        }
    }
}

static int lg_(long n) { int k = 0; while (n > 1) { n >>= 1; ++k; } return k; }

static void introselect_(pair_t *f, long first, long nth, long last, int depth) {
    while (last - first > 3) {
        if (depth == 0) {
            heap_select_(f + first, nth + 1 - first, last - first);
            swap_(&f[first], &f[nth]);
            return;
        }
        --depth;
        long cut = partition_pivot_(f, first, last);
        if (cut <= nth) first = cut; else last = cut;
    }
    insertion_sort_(f, first, last);
}

static void introsort_loop_(pair_t *f, long first, long last, int depth) {
    while (last - first > 16) {
        if (depth == 0) {
            heap_select_(f + first, last - first, last - first);
            sort_heap_(f + first, last - first);
            return;
        }
        --depth;
        long cut = partition_pivot_(f, first, last);
        introsort_loop_(f, cut, last, depth);
        last = cut;
    }
}

static void sort_(pair_t *f, long first, long last) {
    if (first == last) return;
    introsort_loop_(f, first, last, 2 * lg_(last - first));
    if (last - first > 16) {
        insertion_sort_(f, first, first + 16);
        for (long i = first + 16; i != last; ++i) unguarded_linear_insert_(f, i);
    } else {
        insertion_sort_(f, first, last);
    }
}

static void topk_row_(pair_t *q, long n, long k) {
    if (k <= 0) return;
    if (k * 64 <= n) {                       /* std::partial_sort */
        heap_select_(q, k, n);
        sort_heap_(q, k);
    } else {                                 /* std::nth_element + std::sort(first k-1) */
        introselect_(q, 0, k - 1, n, 2 * lg_(n));
        sort_(q, 0, k - 1);
    }
}

/* out_idx: [nrows][k] int64, in torch.topk's output order. Returns 0, or -1 on bad args. */
int editor_oracle_topk_f32(const float *x, long nrows, long n, long k, int64_t *out_idx) {
    if (k > n || n <= 0) return -1;
    pair_t *q = (pair_t *)malloc((size_t)n * sizeof(pair_t));
    if (!q) return -1;
    for (long r = 0; r < nrows; ++r) {
        for (long j = 0; j < n; ++j) {
            float v = x[r * n + j];
            q[j].v = (double)v; q[j].i = j; q[j].nan = isnan(v) ? 1 : 0;
        }
        topk_row_(q, n, k);
        for (long j = 0; j < k; ++j) out_idx[r * k + j] = q[j].i;
    }
    free(q);
    return 0;
}

int editor_oracle_topk_i32(const int32_t *x, long nrows, long n, long k, int64_t *out_idx) {
    if (k > n || n <= 0) return -1;
    pair_t *q = (pair_t *)malloc((size_t)n * sizeof(pair_t));
    if (!q) return -1;
    for (long r = 0; r < nrows; ++r) {
        for (long j = 0; j < n; ++j) { q[j].v = (double)x[r * n + j]; q[j].i = j; q[j].nan = 0; }
        topk_row_(q, n, k);
        for (long j = 0; j < k; ++j) out_idx[r * k + j] = q[j].i;
    }
    free(q);
    return 0;
}
