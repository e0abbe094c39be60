[("1 stream, 2^23, head", dict(NGSLD_TEST_RUN_STREAMS="1")),
 ("1 stream, 2^23, no head", dict(NGSLD_TEST_RUN_STREAMS="1", NGSLD_HEAD="0")),
 ("1 stream, 2^24, head", dict(NGSLD_TEST_RUN_STREAMS="1", NGSLD_TEST_BATCH_PAIRS=str(1 << 24))),
 ("2 streams, 2^22, head", dict(NGSLD_TEST_RUN_STREAMS="2", NGSLD_TEST_BATCH_PAIRS=str(1 << 22))),
 ("1 stream, 2^23, head, tail 1", dict(NGSLD_TEST_RUN_STREAMS="1", NGSLD_TEST_TAIL_LEN="1")),
 ("1 stream, 2^22, head", dict(NGSLD_TEST_RUN_STREAMS="1", NGSLD_TEST_BATCH_PAIRS=str(1 << 22)))]
