from pytorch_distributed_training_amd.parallel import plan_buckets

MiB = 1024 * 1024


def test_reverse_order_small_first_bucket_and_caps():
    numels = [300_000] * 10  # 1.2 MB each in fp32
    plan = plan_buckets(numels, [4] * 10, first_bucket_bytes=1 * MiB, bucket_bytes=4 * MiB)
    # last registered parameter is launched first -> bucket 0
    assert plan.bucket_of[9] == 0
    # bucket ids are non-decreasing when walking parameters in reverse registration order
    rev = [plan.bucket_of[i] for i in reversed(range(10))]
    assert rev == sorted(rev)
    # a parameter larger than the first cap gets bucket 0 to itself
    assert rev.count(0) == 1
    for b, nbytes in enumerate(plan.bucket_bytes):
        members = [i for i in range(10) if plan.bucket_of[i] == b]
        assert len(members) >= 1
        if len(members) > 1:
            assert nbytes <= 4 * MiB
    assert sum(1 for _ in plan.bucket_bytes) == plan.n_buckets


def test_offsets_aligned_and_non_overlapping():
    numels = [7, 130, 64, 1, 1000]
    plan = plan_buckets(numels, [4] * 5, first_bucket_bytes=1 << 30, bucket_bytes=1 << 30, align_bytes=256)
    assert plan.n_buckets == 1
    spans = sorted((plan.offset_bytes[i], plan.offset_bytes[i] + numels[i] * 4) for i in range(5))
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert all(o % 256 == 0 for o in plan.offset_bytes)
    assert plan.bucket_bytes[0] % 256 == 0 and plan.bucket_bytes[0] >= spans[-1][1]


def test_buckets_do_not_mix_element_sizes():
    plan = plan_buckets([100, 100, 100, 100], [4, 2, 2, 4], first_bucket_bytes=1 << 20, bucket_bytes=1 << 20)
    by_bucket = {}
    for i, b in enumerate(plan.bucket_of):
        by_bucket.setdefault(b, set()).add([4, 2, 2, 4][i])
    assert all(len(s) == 1 for s in by_bucket.values())
    assert plan.n_buckets == 3


def test_custom_order():
    plan = plan_buckets([10, 10, 10], [4, 4, 4], first_bucket_bytes=64, bucket_bytes=64, order=[1, 0, 2], align_bytes=1)
    assert plan.bucket_of[1] == 0 and plan.bucket_of[0] == 1 and plan.bucket_of[2] == 2


def test_resnet50_bucket_count_is_small():
    from pytorch_distributed_training_amd.models import build_model

    m = build_model("resnet50")
    ps = [p for p in m.parameters()]
    plan = plan_buckets([p.numel() for p in ps], [4] * len(ps))
    total = sum(p.numel() for p in ps) * 4
    assert total == 25_557_032 * 4
    assert 4 <= plan.n_buckets <= 8


def test_small_final_bucket_holds_the_parameters_that_become_ready_last():
    numels = [1000, 2000, 500_000, 500_000, 500_000, 500_000]  # registration order: the small tensors come first
    plan = plan_buckets(numels, [4] * 6, first_bucket_bytes=1 * MiB, bucket_bytes=8 * MiB, last_bucket_bytes=1 * MiB)
    last = plan.n_buckets - 1
    # the two small, first-registered tensors (ready last in backward) sit alone in the final bucket
    assert [i for i in range(6) if plan.bucket_of[i] == last] == [0, 1]
    assert plan.bucket_bytes[last] <= 1 * MiB
    # without the option they share the bucket of their neighbours
    plain = plan_buckets(numels, [4] * 6, first_bucket_bytes=1 * MiB, bucket_bytes=8 * MiB)
    assert plain.bucket_of[0] == plain.bucket_of[2]
    # bucket ids still never decrease along the visiting order
    rev = [plan.bucket_of[i] for i in reversed(range(6))]
    assert rev == sorted(rev)


def test_final_bucket_never_swallows_everything_or_a_huge_tensor():
    plan = plan_buckets([10, 10], [4, 4], first_bucket_bytes=1 << 20, bucket_bytes=1 << 20, last_bucket_bytes=1 << 20)
    assert plan.n_buckets == 2  # the suffix stops before the first visited parameter
    plan = plan_buckets([4_000_000, 10], [4, 4], first_bucket_bytes=64 << 20, bucket_bytes=64 << 20, last_bucket_bytes=1 << 20)
    assert plan.n_buckets == 1  # the last-ready tensor alone exceeds the final capacity: nothing is split off


def test_final_bucket_respects_element_size_boundaries():
    plan = plan_buckets([100, 100, 100], [2, 4, 4], first_bucket_bytes=1 << 20, bucket_bytes=1 << 20, last_bucket_bytes=1 << 20)
    assert plan.bucket_of[0] != plan.bucket_of[1]
    assert plan.offset_bytes[0] == 0
