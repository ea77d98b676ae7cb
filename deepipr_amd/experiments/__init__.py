"""Train / test loops, kwargs builders and the experiment runner (host mirror of the reference's `experiments`)."""
