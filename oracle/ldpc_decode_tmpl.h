/* ldpc_decode_tmpl.h -- body of decode_ms<T>, hard_to_llrs<T>, llrs_to_hard<T>.
 *
 * TEST INFRASTRUCTURE (see ldpc_oracle.h).  Included once per LLR type by
 * ldpc_decode.c with these macros defined:
 *     T            element type
 *     SUF          symbol suffix (i8, i16, i32, f32, f64)
 *     T_MAX        T::maxval()                 src/decoder.rs:45,54,63,72,81
 *     T_ABS(x)     DecodeFrom::abs             src/decoder.rs:46,55,64,73,82
 *     T_ADD(a,b)   DecodeFrom::saturating_add  src/decoder.rs:47,56,65,74,83
 *     T_SUB(a,b)   DecodeFrom::saturating_sub  src/decoder.rs:48,57,66,75,84
 * hard_bit is `x < 0` for every T (src/decoder.rs:49,58,67,76,85).
 *
 * The statements below follow src/decoder.rs:347-475 one for one; comments give
 * the line each statement restates.
 */
#define CAT_(a, b) a##b
#define CAT(a, b)  CAT_(a, b)

int CAT(oracle_decode_ms_, SUF)(int code, const T *llrs, uint8_t *output, T *working,
                                uint8_t *working_u8, size_t maxiters, size_t *iters_run)
{
    const struct edge_table *tab = oracle_internal_edges(code);
    if (!tab) return -1;

    const size_t n = oracle_code_n(code);                         /* :352 */
    const size_t k = oracle_code_k(code);                         /* :353 */
    const size_t p = oracle_code_punctured_bits(code);            /* :354 */
    const size_t E = tab->n_edges;
    const size_t n_checks = n + p - k;
    const size_t out_len = (n + p) / 8;
    const uint16_t *echk = tab->check, *evar = tab->var;

    uint8_t *parities = output;                                   /* :363 */
    uint8_t *ui_sgns = working_u8;                                /* :367 */
    memset(ui_sgns, 0, n_checks / 8);                             /* :368 */

    const size_t wlen = 2 * E + 3 * n + 3 * p - 2 * k;
    for (size_t i = 0; i < wlen; i++) working[i] = (T)0;          /* :374 */
    T *u       = working;                                         /* :375 */
    T *v       = u + E;                                           /* :376 */
    T *va      = v + E;                                           /* :377 */
    T *ui_min1 = va + (n + p);                                    /* :378 */
    T *ui_min2 = ui_min1 + n_checks;

    int success = 0;
    size_t iters = maxiters;                                      /* :474 */

    for (size_t iter = 0; iter < maxiters; iter++) {              /* :380 */
        memcpy(va, llrs, n * sizeof(T));                          /* :382 */
        for (size_t i = n; i < n + p; i++) va[i] = (T)0;          /* :383 */

        /* pass 1: check-to-variable messages and marginals, :388-411 */
        for (size_t idx = 0; idx < E; idx++) {
            const size_t check = echk[idx], var = evar[idx];
            if (T_ABS(v[idx]) == ui_min1[check])                  /* :391 */
                u[idx] = ui_min2[check];                          /* :392 */
            else
                u[idx] = ui_min1[check];                          /* :394 */
            if ((ui_sgns[check / 8] >> (check % 8)) & 1)          /* :398 */
                u[idx] = -u[idx];                                 /* :399 */
            if (v[idx] < (T)0)                                    /* :403 */
                u[idx] = -u[idx];                                 /* :404 */
            va[var] = T_ADD(va[var], u[idx]);                     /* :408 */
        }

        for (size_t i = 0; i < n_checks; i++) ui_min1[i] = T_MAX; /* :414 */
        for (size_t i = 0; i < n_checks; i++) ui_min2[i] = T_MAX; /* :415 */
        memset(ui_sgns, 0, n_checks / 8);                         /* :416 */
        memset(parities, 0, out_len);                             /* :417 */

        /* pass 2: variable-to-check messages with self-correction, :419-450 */
        for (size_t idx = 0; idx < E; idx++) {
            const size_t check = echk[idx], var = evar[idx];
            const T new_v_ai = T_SUB(va[var], u[idx]);            /* :421 */
            if (((new_v_ai < (T)0) == (v[idx] < (T)0)) || v[idx] == (T)0) /* :422 */
                v[idx] = new_v_ai;                                /* :423 */
            else
                v[idx] = (T)0;                                    /* :425 */

            const T a = T_ABS(v[idx]);
            if (a < ui_min1[check]) {                             /* :430 */
                ui_min2[check] = ui_min1[check];                  /* :431 */
                ui_min1[check] = a;                               /* :432 */
            } else if (a < ui_min2[check]) {                      /* :433 */
                ui_min2[check] = a;                               /* :434 */
            }
            if (v[idx] < (T)0)                                    /* :439 */
                ui_sgns[check / 8] ^= (uint8_t)(1u << (check % 8)); /* :440 */
            if (va[var] < (T)0)                                   /* :445 */
                parities[check / 8] ^= (uint8_t)(1u << (check % 8)); /* :446 */
        }

        /* all parity equations satisfied? :453 (max over output_len bytes) */
        uint8_t worst = 0;
        for (size_t i = 0; i < out_len; i++) if (parities[i] > worst) worst = parities[i];
        if (worst == 0) {
            success = 1;                                          /* :462 */
            iters = iter;
            break;
        }
    }

    /* hard decision of the marginals, MSB first: :455-461 and :467-473 */
    memset(output, 0, out_len);
    for (size_t var = 0; var < n + p; var++)
        if (va[var] < (T)0)
            output[var / 8] |= (uint8_t)(1u << (7 - (var % 8)));

    if (iters_run) *iters_run = iters;                            /* capi/src/lib.rs:91-93 */
    return success;
}

/* hard_to_llrs: src/decoder.rs:484-493 */
void CAT(oracle_hard_to_llrs_, SUF)(int code, const uint8_t *input, T *llrs)
{
    const size_t n = oracle_code_n(code);
    const T llr = (T)-1;                                          /* :487 */
    for (size_t idx = 0; idx < n / 8; idx++)
        for (int i = 0; i < 8; i++)
            llrs[idx * 8 + i] = ((input[idx] >> (7 - i)) & 1) ? llr : (T)-llr;   /* :490 */
}

/* llrs_to_hard: src/decoder.rs:498-509 */
void CAT(oracle_llrs_to_hard_, SUF)(int code, const T *llrs, uint8_t *output)
{
    const size_t n = oracle_code_n(code);
    memset(output, 0, n / 8);                                     /* :502 */
    for (size_t i = 0; i < n; i++)
        if (llrs[i] < (T)0)                                       /* :505 */
            output[i / 8] |= (uint8_t)(1u << (7 - (i % 8)));      /* :506 */
}

/* Batched driver: independent frames, one private working area per thread
 * (perftest/src/main.rs:19-21 allocates per trial; :39-45 runs one per core). */
int CAT(oracle_decode_ms_batch_, SUF)(int code, const T *llrs, uint8_t *output, uint32_t *iters,
                                      uint8_t *success, size_t batch, size_t maxiters, int nthreads)
{
    if (!oracle_internal_edges(code)) return -1;
    const size_t n = oracle_code_n(code);
    const size_t out_len = oracle_output_len(code);
    const size_t wlen = oracle_ms_working_len(code);
    const size_t w8len = oracle_ms_working_u8_len(code);
    if (nthreads <= 0) nthreads = omp_get_num_procs();
    int failed = 0;
#pragma omp parallel num_threads(nthreads)
    {
        T *working = malloc(wlen * sizeof(T));
        uint8_t *working_u8 = malloc(w8len);
        if (!working || !working_u8) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (size_t f = 0; f < batch; f++) {
                size_t it = 0;
                int ok = CAT(oracle_decode_ms_, SUF)(code, llrs + f * n, output + f * out_len,
                                                     working, working_u8, maxiters, &it);
                if (iters) iters[f] = (uint32_t)it;
                if (success) success[f] = (uint8_t)(ok == 1);
            }
        }
        free(working);
        free(working_u8);
    }
    return failed ? -1 : nthreads;
}

#undef CAT
#undef CAT_
