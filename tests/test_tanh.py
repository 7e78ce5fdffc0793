"""``tanh`` as a layer activation (Dense, Conv2D, Activation, the graph front end's ``tf.tanh``): forward values and
all gradients against PyTorch autograd, on the CPU reference path and - marked gpu - on the HIP kernels."""
import pytest
import torch

from distributed_tensorflow_amd.layers import Activation, Conv2D, Dense, Flatten, MaxPooling2D
from distributed_tensorflow_amd.models import Sequential
from distributed_tensorflow_amd.ops import autograd as A
from distributed_tensorflow_amd.ops import functional as F
from distributed_tensorflow_amd.ops._native import ACT_TANH


def _mlp(device):
    m = Sequential([Dense(24, input_dim=12, activation="tanh"), Dense(16), Activation("tanh"), Dense(5, activation="sigmoid")])
    m.compile(loss="mean_squared_error", optimizer="sgd", device=device, seed=3)
    return m


def _cnn(device):
    m = Sequential([Conv2D(8, (3, 3), activation="tanh", padding="same", input_shape=(8, 8, 4)), MaxPooling2D((2, 2)),
                    Conv2D(8, (3, 3), activation="tanh"), Flatten(), Dense(3, activation="sigmoid")])
    m.compile(loss="mean_squared_error", optimizer="sgd", device=device, seed=4)
    return m


def _torch_mlp_grads(m, x, y):
    w = {v.name: v.tensor.detach().cpu().double().requires_grad_(True) for v in m.store.trainable}
    names = [v.name for v in m.store.trainable]
    h = torch.tanh(x.double() @ w[names[0]] + w[names[1]])
    h = torch.tanh(h @ w[names[2]] + w[names[3]])
    p = torch.sigmoid(h @ w[names[4]] + w[names[5]])
    ((p - y.double()) ** 2).mean().backward()
    return p.detach(), {n: w[n].grad for n in names}


def _torch_cnn_grads(m, x, y):
    w = {v.name: v.tensor.detach().cpu().double().requires_grad_(True) for v in m.store.trainable}
    n = [v.name for v in m.store.trainable]
    conv = torch.nn.functional.conv2d
    h = torch.tanh(conv(x.double().permute(0, 3, 1, 2), w[n[0]].permute(3, 2, 0, 1), w[n[1]], padding=1))
    h = torch.nn.functional.max_pool2d(h, 2)
    h = torch.tanh(conv(h, w[n[2]].permute(3, 2, 0, 1), w[n[3]]))
    h = h.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    p = torch.sigmoid(h @ w[n[4]] + w[n[5]])
    ((p - y.double()) ** 2).mean().backward()
    return p.detach(), {k: w[k].grad for k in n}


def check_model(build, reference, x, y, device):
    m = build(device)
    want_p, want_g = reference(m, x, y)
    xd, yd = x.to(device), y.to(device)
    p = m.predict(xd, batch_size=x.shape[0])
    torch.testing.assert_close(torch.as_tensor(p).cpu().double(), want_p, rtol=2e-4, atol=2e-5)
    m.compute_gradients(xd, yd)
    for v in m.store.trainable:
        torch.testing.assert_close(v.grad.cpu().double(), want_g[v.name], rtol=2e-3, atol=2e-6)


def _data(shape, outs, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g), torch.rand(shape[0], outs, generator=g)


def test_dense_and_activation_layers_cpu():
    x, y = _data((40, 12), 5, 1)
    check_model(_mlp, _torch_mlp_grads, x, y, "cpu")


def test_convolution_cpu():
    x, y = _data((6, 8, 8, 4), 3, 2)
    check_model(_cnn, _torch_cnn_grads, x, y, "cpu")


@pytest.mark.gpu
def test_dense_and_activation_layers_gpu(gpu_device):
    x, y = _data((40, 12), 5, 1)
    check_model(_mlp, _torch_mlp_grads, x, y, gpu_device)


@pytest.mark.gpu
def test_convolution_gpu(gpu_device):
    x, y = _data((6, 8, 8, 4), 3, 2)
    check_model(_cnn, _torch_cnn_grads, x, y, gpu_device)


@pytest.mark.gpu
def test_kernels_gpu(gpu_device):
    g = torch.Generator().manual_seed(5)
    x, dy = torch.randn(1000, 37, generator=g) * 3, torch.randn(1000, 37, generator=g)
    yk = F.activation_forward(x.to(gpu_device), ACT_TANH)
    torch.testing.assert_close(yk.cpu(), torch.tanh(x), rtol=1e-5, atol=1e-6)
    dz = F.act_grad(dy.to(gpu_device), yk, ACT_TANH)
    torch.testing.assert_close(dz.cpu(), dy * (1 - torch.tanh(x) ** 2), rtol=1e-4, atol=1e-6)
    w, b = torch.randn(37, 64, generator=g) * 0.2, torch.randn(64, generator=g)
    out = F.dense_forward(x.to(gpu_device), w.to(gpu_device), b.to(gpu_device), ACT_TANH)
    torch.testing.assert_close(out.cpu(), torch.tanh(x @ w + b), rtol=1e-4, atol=1e-5)


def test_autograd_function_and_graph_front_end():
    g = torch.Generator().manual_seed(6)
    x = torch.randn(9, 7, generator=g)
    w = torch.randn(7, 4, generator=g).requires_grad_(True)
    b = torch.randn(4, generator=g).requires_grad_(True)
    A.dense(x, w, b, "tanh").sum().backward()
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    torch.tanh(x @ w2 + b2).sum().backward()
    torch.testing.assert_close(w.grad, w2.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(b.grad, b2.grad, rtol=1e-5, atol=1e-6)

    from distributed_tensorflow_amd import compat as tf
    tf.reset_default_graph()
    xin = tf.placeholder(tf.float32, shape=(None, 7))
    lab = tf.placeholder(tf.float32, shape=(None, 3))
    W1 = tf.Variable(tf.random_normal([7, 10], stddev=0.3), name="W1")
    b1 = tf.Variable(tf.zeros([10]), name="b1")
    h = tf.tanh(tf.matmul(xin, W1) + b1)
    out = tf.layers.dense(h, 3, activation=tf.nn.tanh)
    loss = tf.reduce_mean(tf.square(out - lab))
    train = tf.train.AdagradOptimizer(0.1).minimize(loss)
    with tf.Session() as sess:
        sess.run(tf.global_variables_initializer())
        feed = {xin: torch.randn(20, 7, generator=g).numpy(), lab: (torch.rand(20, 3, generator=g) - 0.5).numpy()}
        first = sess.run(loss, feed)
        for _ in range(30):
            sess.run(train, feed)
        assert sess.run(loss, feed) < first
        model = tf.get_default_graph().engine().model
    assert [l.activation for l in model.layers if isinstance(l, Dense)] == ["tanh", "tanh"]
    assert model.optimizer.name == "Adagrad"
